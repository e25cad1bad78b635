// sdt_scatter_seq_d.hip -- instantiations of the one-lane-per-read level-1 scatter (sdt_sk_scatter_seq.cuh), compiled on their own
#include "sdt_sk_scatter_seq.cuh"

hipError_t sk_seq_launch_nw2_hi(int w, const SkSeqLaunch &a, const Table<2> &tbl)
{
	switch (w) {
	case 45: return sk_seq_launch_one<2, 45>(a, tbl);
	case 47: return sk_seq_launch_one<2, 47>(a, tbl);
	case 49: return sk_seq_launch_one<2, 49>(a, tbl);
	case 51: return sk_seq_launch_one<2, 51>(a, tbl);
	case 53: return sk_seq_launch_one<2, 53>(a, tbl);
	default: return hipErrorInvalidValue;
	}
}
