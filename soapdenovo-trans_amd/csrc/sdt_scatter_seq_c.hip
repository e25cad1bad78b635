// sdt_scatter_seq_c.hip -- instantiations of the one-lane-per-read level-1 scatter (sdt_sk_scatter_seq.cuh), compiled on their own
#include "sdt_sk_scatter_seq.cuh"

hipError_t sk_seq_launch_nw2_mid(int w, const SkSeqLaunch &a, const Table<2> &tbl)
{
	switch (w) {
	case 35: return sk_seq_launch_one<2, 35>(a, tbl);
	case 37: return sk_seq_launch_one<2, 37>(a, tbl);
	case 39: return sk_seq_launch_one<2, 39>(a, tbl);
	case 41: return sk_seq_launch_one<2, 41>(a, tbl);
	case 43: return sk_seq_launch_one<2, 43>(a, tbl);
	default: return hipErrorInvalidValue;
	}
}
