/* libcfg.c -- see libcfg.h.  Format facts reproduced (all from lib.c):
 *  - a line whose first five characters are "[LIB]" opens a library (:129-136,184-190);
 *  - every other line is split into two tokens: maximal runs of printable ASCII (32..126) other than '='
 *    (:58-97) -- so blanks are PART of a token and "avg_ins = 200" is not recognised, exactly as upstream;
 *  - before the first [LIB] only `max_rd_len` is looked at (:138-152);
 *  - keys: f1 f2 q1 q2 f q p b, min_ins max_ins avg_ins rd_len_cutoff reverse_seq asm_flags rank
 *    pair_num_cutoff map_len (:354-433); defaults asm_flag=3, everything else 0 (:156-171);
 *  - libraries are sorted by avg_ins (:437). */
#include "libcfg.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int split2(const char *line, char tok[2][1024])
{
	int n = 0;
	size_t i = 0, len = strlen(line);
	while (i < len && n < 2) {
		unsigned char c = (unsigned char)line[i];
		if (c >= 32 && c <= 126 && c != '=') {
			size_t j = 0;
			while (i < len && (unsigned char)line[i] >= 32 && (unsigned char)line[i] <= 126 && line[i] != '=') {
				if (j < 1023) tok[n][j++] = line[i];
				i++;
			}
			tok[n][j] = '\0';
			n++;
		} else {
			i++;
		}
	}
	return n == 2;
}

static void push_name(char ***list, int *n, const char *name)
{
	*list = (char **)realloc(*list, (size_t)(*n + 1) * sizeof(char *));
	(*list)[*n] = strdup(name);
	(*n)++;
}

static int cmp_lib(const void *a, const void *b)
{
	const sdt_lib *A = (const sdt_lib *)a, *B = (const sdt_lib *)b;
	if (A->avg_ins != B->avg_ins) return A->avg_ins > B->avg_ins ? 1 : -1;
	return A->order - B->order;          /* qsort is not stable upstream; ties keep file order here */
}

int sdt_cfg_load(const char *path, sdt_cfg *cfg)
{
	FILE *fp = fopen(path, "r");
	char line[1024], tok[2][1024];
	memset(cfg, 0, sizeof *cfg);
	if (!fp) {
		printf("Cannot open %s. Now exit to system...\n", path);       /* check.c:26-35 wording */
		return -1;
	}
	sdt_lib *cur = NULL;
	while (fgets(line, sizeof line, fp)) {
		if (strncmp(line, "[LIB]", 5) == 0) {
			cfg->libs = (sdt_lib *)realloc(cfg->libs, (size_t)(cfg->nlibs + 1) * sizeof(sdt_lib));
			cur = &cfg->libs[cfg->nlibs];
			memset(cur, 0, sizeof *cur);
			cur->asm_flag = 3;
			cur->order = cfg->nlibs++;
			continue;
		}
		if (!split2(line, tok))
			continue;
		if (!cur) {
			if (strcmp(tok[0], "max_rd_len") == 0)
				cfg->max_rd_len = atoi(tok[1]);
			continue;
		}
		const char *k = tok[0], *v = tok[1];
		if (!strcmp(k, "f1")) push_name(&cur->f1, &cur->nf1, v);
		else if (!strcmp(k, "f2")) push_name(&cur->f2, &cur->nf2, v);
		else if (!strcmp(k, "q1")) push_name(&cur->q1, &cur->nq1, v);
		else if (!strcmp(k, "q2")) push_name(&cur->q2, &cur->nq2, v);
		else if (!strcmp(k, "f")) push_name(&cur->f, &cur->nf, v);
		else if (!strcmp(k, "q")) push_name(&cur->q, &cur->nq, v);
		else if (!strcmp(k, "p")) push_name(&cur->p, &cur->np, v);
		else if (!strcmp(k, "b")) push_name(&cur->b, &cur->nb, v);
		else if (!strcmp(k, "min_ins")) cur->min_ins = atoi(v);
		else if (!strcmp(k, "max_ins")) cur->max_ins = atoi(v);
		else if (!strcmp(k, "avg_ins")) cur->avg_ins = atoi(v);
		else if (!strcmp(k, "rd_len_cutoff")) cur->rd_len_cutoff = atoi(v);
		else if (!strcmp(k, "reverse_seq")) cur->reverse = atoi(v);
		else if (!strcmp(k, "asm_flags")) cur->asm_flag = atoi(v);
		else if (!strcmp(k, "rank")) cur->rank = atoi(v);
		else if (!strcmp(k, "pair_num_cutoff")) cur->pair_num_cut = atoi(v);
		else if (!strcmp(k, "map_len")) cur->map_len = atoi(v);
	}
	fclose(fp);
	if (cfg->nlibs > 1)
		qsort(cfg->libs, (size_t)cfg->nlibs, sizeof(sdt_lib), cmp_lib);
	return 0;
}

static void free_list(char **l, int n)
{
	for (int i = 0; i < n; i++) free(l[i]);
	free(l);
}

void sdt_cfg_free(sdt_cfg *cfg)
{
	for (int i = 0; i < cfg->nlibs; i++) {
		sdt_lib *l = &cfg->libs[i];
		free_list(l->f1, l->nf1); free_list(l->f2, l->nf2); free_list(l->q1, l->nq1); free_list(l->q2, l->nq2);
		free_list(l->p, l->np); free_list(l->b, l->nb); free_list(l->f, l->nf); free_list(l->q, l->nq);
	}
	free(cfg->libs);
	memset(cfg, 0, sizeof *cfg);
}
