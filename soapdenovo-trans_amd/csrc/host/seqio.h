/* seqio.h -- read ingest for the MI355X pregraph host: mmap + parallel parse of FASTQ / FASTA straight
 * into the packed 2-bit stream of include/sdt_gpu.h.
 *
 * Reference semantics kept (readseq1by1.c:122-178 readseqInBuf, :281-340 readseqfq): the sequence line is
 * cut to max_read_len characters first; then lowercase is folded, letters are coded (c & 6) >> 1 (A0 C1 T2
 * G3, N and every other letter by their bits), '.' is A, anything else is dropped; reverse_seq libraries
 * reverse-complement every read (reverse2k :749-764).  NOT kept: the 32 KiB POSIX-AIO chunker
 * (prlHashReads.c:718-806) and its hang on files that are a multiple of 32768 bytes. */
#ifndef SDT_SEQIO_H
#define SDT_SEQIO_H
#include <stdint.h>
#include <stddef.h>

typedef struct {
	uint32_t *words;        /* packed stream, 4 zero pad words at the end */
	uint64_t nwords;
	uint64_t *offsets;      /* nreads + 1, in bases */
	uint64_t nreads;        /* every record of the chunk, also reads shorter than K+1 (the device skips them) */
	int owner;              /* multi-process runs (sdt_read_shard): the rank that counts this chunk in pass 1 */
	int counted_only;       /* 1: the chunk belongs to another rank and was only counted (words == offsets == NULL) */
	int pool_slot;          /* >= 0: words / offsets live in a buffer of the pool below (sdt_pool_take to keep it past the callback) */
	uint64_t fixed_len;     /* > 0: every read of the batch has exactly this many bases (offsets[i] = i * fixed_len) */
	uint64_t text_bytes;    /* bytes of the input file this batch was parsed from */
	/* multi-process runs: where the chunk stands in the pass (readstream.c fills the stream fields) */
	uint64_t chunk_index;   /* number of the chunk over all files of the pass (owner = chunk_index % nranks) */
	int count_unknown;      /* 1: a foreign chunk that was not even scanned (sdt_read_shard_skip_foreign): nreads is 0, the owner knows */
	int stream_id;          /* number of the file within the pass */
	int stream_parity;      /* paired files: 0 = the reads of this file take the even ordinals of the pair, 1 = the odd ones */
} sdt_batch;

/* Optional pool of output buffers for a host that hands batches to the device ASYNCHRONOUSLY from pinned memory (the
 * reference double-buffers its read buffers the same way, prlHashReads.c:493-620): chunks are packed straight into pool
 * buffers -- allocated through `alloc` (pinned memory behind the C ABI), sized by the chunk length -- instead of fresh
 * malloc blocks.  A worker takes its buffer together with its chunk number, so the chunk the consumer waits for always has
 * one.  The consumer's callback may keep a batch's buffer after it returns (sdt_pool_take) and gives it back when the
 * copy has left it (sdt_pool_release); otherwise the reader recycles it.  sdt_pool_disable frees everything. */
void sdt_pool_enable(void *(*alloc)(size_t bytes), void (*release)(void *p), int nslots);
void sdt_pool_disable(void);
void sdt_pool_take(int slot);
void sdt_pool_release(int slot);

/* Multi-process runs (`sdt-pregraph --gpus N`): every rank walks the same chunks in the same order, chunk i (counted
 * over all files of a pass) belongs to rank i % nranks.  A rank only COUNTS the records of foreign chunks (the read
 * ordinals of everything after them depend on it) unless keep_all is set (rank 0 keeps every read for the second
 * pass).  sdt_read_shard_begin resets the chunk counter: call it before each pass over the reads.
 * sdt_read_shard_skip_foreign(1): a rank does not look at foreign chunks at all -- it touches the text around the chunk
 * boundaries (the cut points are found the same way on every rank) and its own chunks, nothing else; such a batch comes with
 * count_unknown = 1, and the caller gets the record counts from the owners (sdt_pregraph.c: one all-gather per group of nranks
 * chunks) and works the ordinals out itself (readstream.h: sdt_stream_ordinals).  The reference has ONE reader for all threads
 * (prlHashReads.c:432-620); N ranks that each scan the whole input would be N readers of 64 GB of text. */
void sdt_read_shard_begin(int rank, int nranks, int keep_all);
void sdt_read_shard_skip_foreign(int on);
/* bytes of text this process parsed / was shown (own chunks / all chunks), summed over all files since the last shard_begin */
extern uint64_t sdt_reader_bytes_parsed, sdt_reader_bytes_seen;

typedef int (*sdt_batch_fn)(void *user, const sdt_batch *b);
/* where the consumer thread's time went, summed over all files (measurement) */
extern double sdt_reader_wait_ms, sdt_reader_fn_ms;

/* Parse one file with `threads` workers; call `fn` on the calling thread for each batch in file order.
 * fmt: 'q' FASTQ, 'a' FASTA.  Returns 0, or -1 after printing a message. */
int sdt_read_file(const char *path, int fmt, int max_read_len, int reverse, int threads, size_t chunk_bytes,
                  sdt_batch_fn fn, void *user, uint64_t *nreads_out);

#endif
