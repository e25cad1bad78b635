// sdt_scatter_seq_b.hip -- instantiations of the one-lane-per-read level-1 scatter (sdt_sk_scatter_seq.cuh), compiled on their own
#include "sdt_sk_scatter_seq.cuh"

hipError_t sk_seq_launch_nw2_lo(int w, const SkSeqLaunch &a, const Table<2> &tbl)
{
	switch (w) {
	case 23: return sk_seq_launch_one<2, 23>(a, tbl);
	case 25: return sk_seq_launch_one<2, 25>(a, tbl);
	case 27: return sk_seq_launch_one<2, 27>(a, tbl);
	case 29: return sk_seq_launch_one<2, 29>(a, tbl);
	case 31: return sk_seq_launch_one<2, 31>(a, tbl);
	case 33: return sk_seq_launch_one<2, 33>(a, tbl);
	default: return hipErrorInvalidValue;
	}
}
