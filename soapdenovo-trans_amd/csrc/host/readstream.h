/* readstream.h -- walk every read of a library config in the reference's consumption order
 * (libraries sorted by avg_ins, lib.c:437; only asm_flags 1 or 3, readseq1by1.c:563; per library f1/f2 pairs,
 * q1/q2 pairs, p, f, q, readseq1by1.c:579-632; paired files alternate read1, read2, prlHashReads.c:493-567).
 * Both passes over the reads (hashing on the GPU, read -> edge paths on the host) use it, so they agree on
 * read ordinals: batch read i has ordinal ord_base + i * ord_stride. */
#ifndef SDT_READSTREAM_H
#define SDT_READSTREAM_H
#include "libcfg.h"
#include "seqio.h"

typedef int (*sdt_stream_fn)(void *user, const sdt_batch *b, uint64_t ord_base, uint64_t ord_stride);

/* Ordinals for a caller that cannot take them from the callback: with sdt_read_shard_skip_foreign a rank does not know how many
 * records the foreign chunks hold until their owners say so, and every ordinal behind such a chunk depends on it.  Feed the chunks
 * of the pass IN ORDER (stream id, stride and parity as the callback reported them, the record count from whoever parsed the chunk):
 * _next returns the ordinal of the chunk's first read -- the value sdt_stream_reads itself would have passed as ord_base. */
typedef struct { uint64_t ordinal, n_cur, n_first; int sid, stride, parity, open_pair; } sdt_stream_ordinals;
void sdt_stream_ordinals_init(sdt_stream_ordinals *s);
uint64_t sdt_stream_ordinals_next(sdt_stream_ordinals *s, int stream_id, int stride, int parity, uint64_t nreads);

/* returns 0, or -1 after printing a message; *nreads = records seen */
int sdt_stream_reads(const sdt_cfg *cfg, int max_read_len, int threads, size_t chunk_bytes, int verbose,
                     sdt_stream_fn fn, void *user, uint64_t *nreads);

#endif
