// sdt_scatter_seq_a.hip -- instantiations of the one-lane-per-read level-1 scatter (sdt_sk_scatter_seq.cuh), compiled on their own
#include "sdt_sk_scatter_seq.cuh"

hipError_t sk_seq_launch_nw1(int w, const SkSeqLaunch &a, const Table<1> &tbl)
{
	switch (w) {
	case 9: return sk_seq_launch_one<1, 9>(a, tbl);
	case 11: return sk_seq_launch_one<1, 11>(a, tbl);
	case 13: return sk_seq_launch_one<1, 13>(a, tbl);
	case 15: return sk_seq_launch_one<1, 15>(a, tbl);
	case 17: return sk_seq_launch_one<1, 17>(a, tbl);
	case 19: return sk_seq_launch_one<1, 19>(a, tbl);
	case 21: return sk_seq_launch_one<1, 21>(a, tbl);
	default: return hipErrorInvalidValue;
	}
}
