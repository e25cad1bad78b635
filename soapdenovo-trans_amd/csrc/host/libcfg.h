/* libcfg.h -- library config file of `pregraph -s <cfg>` (reference: lib.c:118-438 scan_libInfo,
 * README.md:117-147).  Written from the format's behaviour, not from the reference's code. */
#ifndef SDT_LIBCFG_H
#define SDT_LIBCFG_H

typedef struct {
	int avg_ins, min_ins, max_ins, reverse, asm_flag, rank, pair_num_cut, rd_len_cutoff, map_len;
	char **f1, **f2, **q1, **q2, **p, **b, **f, **q;     /* file name lists, in file order */
	int nf1, nf2, nq1, nq2, np, nb, nf, nq;
	int order;                                             /* position in the file (stable sort key) */
} sdt_lib;

typedef struct {
	int max_rd_len;        /* 0 when absent: the caller applies the reference default of 100 */
	int nlibs;
	sdt_lib *libs;         /* sorted by avg_ins ascending (lib.c:437) */
} sdt_cfg;

/* returns 0 on success; on failure prints a message and returns -1 */
int sdt_cfg_load(const char *path, sdt_cfg *cfg);
void sdt_cfg_free(sdt_cfg *cfg);

#endif
