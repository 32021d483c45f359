/*
 * lbzip2_amd.h -- C ABI of the MI355X-native bzip2 block-compression core.
 *
 * Two faces of the same engine (lbzip2_amd/csrc, hand-written HIP for gfx950):
 *
 * (A) THE DROP-IN WORK-UNIT INTERFACE.  Exactly the functions lbzip2's pipeline
 *     (src/compress.c:89-94, 113, 220-223) calls, with the signatures and semantics of the
 *     reference's src/encode.h:22-38, so that compress.c/process.c can be linked against this
 *     library instead of encode.o + divbwt.o + crctab.o and drive GPU blocks instead of
 *     pthread CPU blocks:
 *
 *       encoder_alloc_size()  <- encode.h:29 / encode.c:108-114
 *       encoder_init()        <- encode.h:30 / encode.c:117-132
 *       collect()             <- encode.h:31 / encode.c:135-336
 *       encode()              <- encode.h:32 / encode.c:427-545
 *       transmit()            <- encode.h:33 / encode.c:1152-1281
 *       combine_crc()         <- encode.h:38
 *
 *     The same entry points are also exported with an lbzamd_ prefix for hosts that cannot
 *     afford such generic symbol names (ctypes, cgo, JNI).
 *
 * (B) THE BATCH INTERFACE.  One call compresses a whole buffer: the slab split of
 *     process.c:631, the work-unit loop of compress.c:73-118, the in-order mux and CRC fold of
 *     compress.c:238-250 and the header/trailer of compress.c:291-321 all run on the device,
 *     hundreds of blocks in flight, one bzip2 block per workgroup.
 *
 * All functions are thread-safe on distinct contexts / encoder states.  There is no CPU
 * fallback: if no HIP device is usable every entry point fails (batch: negative return;
 * drop-in: message on stderr + abort(), the reference's own convention for fatal errors,
 * main.c:59-84).
 */
#ifndef LBZIP2_AMD_H
#define LBZIP2_AMD_H

#include <stddef.h>
#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ (A) drop-in */
#define CLUSTER_FACTOR  8u          /* encode.h:22 */
#define HEADER_SIZE     4u          /* encode.h:23 */
#define TRAILER_SIZE    10u         /* encode.h:24 */

struct encoder_state;               /* opaque, caller-allocated: malloc(encoder_alloc_size(mbs)) */

size_t encoder_alloc_size(unsigned long mbs);
void   encoder_init(struct encoder_state *e, unsigned long mbs, unsigned cf);
/* Consumes a prefix of buf[0..*buf_sz) into the block (RLE1 + CRC); *buf_sz = bytes left
 * unconsumed.  Returns 1 if the block filled up before the input ran out.
 * How this implementation serves the two call patterns of compress.c:
 *   - default mode (:73-118): ONE collect() per state on at most max_block_size bytes; the calls of all worker
 *     threads are batched into rounds on the device.  What does not fit is left unconsumed and the return value is
 *     1 ("block full", encode.c:335): the caller re-queues it as the slab's next work unit (compress.c:98-104);
 *   - -u / --sequential (:129-198): collect() called AGAIN on a state that already holds bytes appends to its
 *     block: the block's raw bytes move to a device buffer of their own and the block is tokenised again from its
 *     start with the new input behind it; the return value is 1 once input is left over.  One such call at a time
 *     (the reference holds a token around it); a call on a full block takes nothing and returns 1;
 *   - encode() on a state that never collected a byte is a fatal error (message on stderr + abort(), the
 *     reference's convention for fatal errors).                                                          */
int    collect(struct encoder_state *e, const uint8_t *buf, size_t *buf_sz);
/* Runs BWT, MTF/ZRLE and prefix-code selection; returns the exact compressed size in bytes
 * and the block's un-inverted CRC through *crc.                                          */
size_t encode(struct encoder_state *e, uint32_t *crc);
/* Writes the compressed block (size rounded up to a multiple of 4 bytes) to buf, or to an
 * internal buffer if buf == NULL; returns the buffer.                                    */
void  *transmit(struct encoder_state *e, void *buf);

#define combine_crc(cc, c) (((cc) << 1) ^ ((cc) >> 31) ^ (c) ^ -1)   /* encode.h:38 */

size_t lbzamd_encoder_alloc_size(unsigned long mbs);
void   lbzamd_encoder_init(struct encoder_state *e, unsigned long mbs, unsigned cf);
int    lbzamd_collect(struct encoder_state *e, const uint8_t *buf, size_t *buf_sz);
size_t lbzamd_encode(struct encoder_state *e, uint32_t *crc);
void  *lbzamd_transmit(struct encoder_state *e, void *buf);
/* Give back the device resources of a state that will not reach transmit(). */
void   lbzamd_encoder_abandon(struct encoder_state *e);

/* ------------------------------------------------------------------ (B) batch */
typedef struct lbzamd_ctx lbzamd_ctx;

typedef struct lbzamd_stats {
  uint64_t n_in;        /* input bytes */
  uint64_t n_rle;       /* sum of block lengths after RLE1 */
  uint64_t n_mtf;       /* sum of MTF/ZRLE symbols */
  uint64_t n_out;       /* stream bytes */
  uint64_t sort_elems;  /* elements passed through the radix sorter (all rounds) */
  uint32_t nblocks;
  uint32_t nperiodic;   /* exactly periodic blocks (origin pointer = smallest equal row) */
  /* Device time from HIP events.  ms_total is wall time on the context's stream.  The per-kernel
     figures are sums over launches on the stream each launch went to; rounds of blocks run on three
     streams, so launches overlap and the sums exceed ms_total (LBZAMD_STREAMS=1: no overlap). */
  float ms_collect, ms_bwt, ms_mtf, ms_encode, ms_finish, ms_total;
  float ms_bwt_part, ms_bwt_batch, ms_bwt_fix;   /* the BWT stage's launches: partition, batches, ties (k_bwt_deep's text rounds + the
                                                    rank rounds k_bwt_fix0 / k_bwt_fixr / k_bwt_fixend); sum = ms_bwt */
  uint32_t seq_fast_links;   /* sequential mode: blocks whose start was found through the step tables (the others walked) */
} lbzamd_stats;

/* device < 0: current device.  max_slabs: slabs resident at once (input beyond that is
 * streamed through in chunks).  nslots: slabs per round (a round = one launch of every kernel,
 * one workgroup per block; each round owns one BWT workspace slot per slab); 0 = max_slabs dealt
 * evenly over the streams, at least one per CU, at most half of the free device memory.
 * Environment: LBZAMD_STREAMS (1..8, default 3), LBZAMD_SLOTS (default for nslots = 0).        */
int  lbzamd_create(lbzamd_ctx **ctx, int device, unsigned bs100k, unsigned max_slabs, unsigned nslots);
void lbzamd_destroy(lbzamd_ctx *ctx);
const char *lbzamd_last_error(void);

/* d_in/d_out are device pointers.  Writes a complete .bz2 stream; *out_len = its size.
 * Work is enqueued on the context's stream and waited for.  0 on success.
 * Alignment: none required.  The kernels read aligned 16-byte vectors (the decoder: aligned 4- and 8-byte words); the only
 * bytes outside [d_in, d_in + len) they ever touch lie in the aligned vector that holds the buffer's first or its last
 * byte -- in one page with a byte of the buffer, so no access can fault whatever lies behind the allocation -- and are
 * masked out.  Nothing outside [d_out, d_out + out_cap) is written.                                                  */
/* The reference's -u / --sequential (main.c; compress.c:129-198 do_collect_seq): with on != 0 the context's
 * following calls cut blocks where they are FULL (a continuous RLE1 over the input, bzip2's own blocking) instead
 * of at every bs100k * 100000 input bytes; the stream is byte-identical to `lbzip2 -u`.  A block's start is known
 * only when its predecessor has been cut, so the blocks' first pass runs as a chain on the device; the body-only
 * calls (a slab range of a bigger stream) are refused in this mode.  max_slabs then counts blocks per chunk. */
int  lbzamd_set_sequential(lbzamd_ctx *ctx, int on);
int  lbzamd_compress_device(lbzamd_ctx *ctx, const void *d_in, size_t len,
                            void *d_out, size_t out_cap, size_t *out_len);
/* Multi-GPU shards: the blocks of a slab-aligned range only (no "BZh9", no trailer), plus the
 * range's block count and CRC fold from zero.  A muxer splices ranges in order into ONE stream that
 * is byte-identical to the single-device (and the reference's) stream: header (compress.c:291-302),
 * bodies in range order (compress.c:238-250), trailer with lbzamd_fold_parts() (compress.c:304-321,
 * combine_crc encode.h:38 -- the fold is GF(2)-linear, so 12 bytes per range suffice).        */
typedef struct lbzamd_part {
  uint64_t bytes;       /* body bytes written */
  uint32_t nblocks;     /* blocks in the range */
  uint32_t crc_fold;    /* combine_crc over the range's blocks, started from 0 */
} lbzamd_part;
int  lbzamd_compress_device_body(lbzamd_ctx *ctx, const void *d_in, size_t len,
                                 void *d_out, size_t out_cap, size_t *out_len, lbzamd_part *part);
int  lbzamd_compress_host_body(lbzamd_ctx *ctx, const uint8_t *in, size_t len,
                               uint8_t *out, size_t out_cap, size_t *out_len, lbzamd_part *part);
/* cc after appending the ranges parts[0..nparts) to a stream whose combined CRC so far is cc:
 * cc = rotl32(cc, nblocks mod 32) ^ crc_fold, range by range. */
uint32_t lbzamd_fold_parts(uint32_t cc, const lbzamd_part *parts, size_t nparts);
/* Same with host buffers (H2D + compress + D2H). */
int  lbzamd_compress_host(lbzamd_ctx *ctx, const uint8_t *in, size_t len,
                          uint8_t *out, size_t out_cap, size_t *out_len);
/* Page-locked host buffers for a host-side splitter/muxer (process.c:260-307 reads, :351-417 writes):
 * H2D/D2H of pinned memory run at link rate and overlap with kernels. */
void *lbzamd_pinned_alloc(size_t bytes);
void  lbzamd_pinned_free(void *p);
/* HIP devices of this process (0 if none).  A host program drives several GPUs the way the reference drives its
 * workers (process.c:515-548): one context per device (lbzamd_create's `device`), slab ranges dealt over them, the
 * ranges muxed in order (lbzamd_compress_*_body + lbzamd_fold_parts); pinned buffers are usable from every device.
 * The drop-in symbols do the same behind the caller's back: LBZAMD_DEVICES=N (default 1, "all" = every device) keeps
 * one work-unit pool per device and leases the states' slabs round-robin over them, so the reference's unmodified
 * splitter/muxer (process.c) feeds N GPUs.                                                                       */
int   lbzamd_device_count(void);
/* Upper bound of the stream size for len input bytes. */
size_t lbzamd_bound(size_t len);
int  lbzamd_get_stats(lbzamd_ctx *ctx, lbzamd_stats *st);
/* Slabs per round (see lbzamd_create). */
uint32_t lbzamd_slots(lbzamd_ctx *ctx);
/* Launch geometry of a round of `blocks` blocks (diagnostic; bench.py names it in its workload string): workgroups per
 * block in the sorting kernels (segments) and in the partition (1: k_bwt_part, one workgroup per block; else the
 * launch-per-pass kernels).  overlapped: other rounds run beside it on their own streams.                          */
void lbzamd_round_shape(lbzamd_ctx *ctx, uint32_t blocks, int overlapped, uint32_t *segments, uint32_t *partition_wgs);
/* The HIP stream (hipStream_t) a caller orders against: every call starts and ends on it (rounds
 * fan out to internal side streams and are joined before the call's last kernels).          */
void *lbzamd_stream(lbzamd_ctx *ctx);

/* ------------------------------------------------------------------ (C) the inverse path
 * Block-parallel decompression of complete .bz2 streams (one or several concatenated), the device-side
 * counterpart of the reference's scan() src/parse.c:282, retrieve() src/decode.c:519, decode() :852 and
 * emit() :944.  Every block of the input is decoded at once, one block per workgroup; block CRCs and the
 * stream CRCs are checked.  max_blocks: blocks decoded per pass (more are taken in several passes).      */
typedef struct lbzamd_dctx lbzamd_dctx;
typedef struct lbzamd_dstats {
  uint64_t n_in, n_out;
  uint32_t nblocks, nstreams;
  /* ms_scan, ms_blocks, ms_emit: HIP events around the three launches (ms_total = their sum).  A block's
     codes, sort and walk run back to back in the ms_blocks kernel: ms_huff / ms_sort / ms_walk are the
     SLOWEST block's stage times (device clock), so they need not add up to ms_blocks. */
  float ms_scan, ms_huff, ms_sort, ms_walk, ms_emit, ms_total, ms_blocks;
} lbzamd_dstats;
int  lbzamd_dcreate(lbzamd_dctx **ctx, int device, unsigned max_blocks);
void lbzamd_ddestroy(lbzamd_dctx *ctx);
/* 0 ok; -1 bad argument / HIP error; -2 output buffer too small (*out_len = bytes needed);
 * -3 malformed stream or CRC mismatch (lbzamd_last_error() says which block): *out_len is then the number of bytes IN FRONT
 *    of the error that were decoded and are in place -- the whole blocks before the first one that was refused -- as the
 *    reference has written what it decoded by then (expand.c:735 writes every buffer that reaches the muxer in order).   */
int  lbzamd_decompress_device(lbzamd_dctx *ctx, const void *d_in, size_t len, void *d_out, size_t out_cap, size_t *out_len);
/* For callers that do not know the decoded size: ONE pass, the result in a malloc'ed buffer (*out, release with
 * lbzamd_free).  (lbzamd_decompress_host with out_cap too small returns -2 and the size AFTER decoding every block;
 * calling it again decodes them again.)  Same return codes otherwise; on -3 with bytes in front of the error *out holds them
 * (release it) and *out_len says how many.                                                                         */
int  lbzamd_decompress_alloc(lbzamd_dctx *ctx, const uint8_t *in, size_t len, uint8_t **out, size_t *out_len);
void lbzamd_free(void *p);
/* A stream taken WINDOW BY WINDOW, in bounded memory (the reference reads, decodes and writes as it goes: src/expand.c:547-690,
 * src/process.c:260-307): `in` holds the bytes of the input from the carry position of the window before it (the first
 * window: from the start of the file) up to as far as the caller has read, `final` says that the input ends there.  The whole
 * blocks of the window are decoded -- a block counts as whole when the magic behind it is in the window too -- and come back
 * as lbzamd_decompress_alloc's do; `rs` (zeroed before the first window) carries the parser's state to the next call, and
 * rs->consumed_bit says where it begins: the caller keeps the bytes from consumed_bit / 8 on, reads more behind them and calls
 * again.  consumed_bit / 8 == 0 with final == 0 means the window held no whole block: it must grow.  rs->finished: the last
 * stream is closed and what follows is not a stream header -- the rest of the input is ignored, as bzip2 does.
 * An input that fits one window (first and final at once) is treated exactly as lbzamd_decompress_alloc treats it.  Across
 * windows one thing differs from it: a block-level error (CRC, size, origin pointer) is reported with its window instead of
 * being held back until the parser has looked at everything behind it. */
typedef struct lbzamd_dresume {
  uint64_t consumed_bit;     /* out: bit of `in` the next window starts at (its bits below consumed_bit % 8 are spent) */
  uint32_t started;          /* 0 before the first window */
  uint32_t in_stream, level, cc, stream_blocks, finished, nblocks_total, nstreams_total;
  uint64_t base_bytes;       /* bytes of the input in front of the next window */
} lbzamd_dresume;
int  lbzamd_decompress_window(lbzamd_dctx *ctx, const uint8_t *in, size_t len, int final, lbzamd_dresume *rs, uint8_t **out, size_t *out_len);
int  lbzamd_decompress_host(lbzamd_dctx *ctx, const uint8_t *in, size_t len, uint8_t *out, size_t out_cap, size_t *out_len);
int  lbzamd_dget_stats(lbzamd_dctx *ctx, lbzamd_dstats *st);
/* After a -3: WHY the stream was refused, as the reference's enum error (src/common.h:54-76: 3 ERR_MAGIC bad stream header
 * ... 15 ERR_BLKCRC, 16 ERR_STRMCRC ... 19 ERR_EOF), so that a host program can word it as lbzip2 does
 * (src/expand.c:69-94 err2str; lbzip2_amd/host/lbzamd.c).  0 after anything else.                                    */
int  lbzamd_last_error_code(void);

/* ------------------------------------------------------------------ (D) the inverse path's work-unit interface
 * reference src/decode.h:38-81, verbatim in its types and prototypes: one compressed block of one worker thread --
 * what src/expand.c:547-690 (do_retrieve / do_emit) drives.  Link the reference's expand.c, parse.c and the rest of
 * its program against this library instead of decode.c and `lbzip2 -d` decodes its blocks on the GPU
 * (tests/test_dropin_link.py).
 *   retrieve()  takes the block's bits out of the caller's bitstream (src/decode.c:519).  Where a block ends is known
 *               only once it is decoded, so the bits the caller has are gathered and the block is decoded on the device
 *               (prefix codes, move-to-front, inverse BWT, inverse RLE1: k_dblock, k_demit); if they run out before the
 *               block does, MORE is returned as the reference does and the next call carries on with more input.  On
 *               OK the bitstream stands behind the block's last code.
 *   decode()    (src/decode.c:852: the counting sort of the inverse BWT) has nothing left to do.
 *   emit()      hands the decoded bytes out, as many as the caller's buffer takes per call (MORE until the last), and
 *               leaves the block's CRC in ds->crc (src/decode.c:944-1146).
 * Error values are the reference's enum error (src/common.h:54-76).                                                  */
struct in_blk;
struct bitstream {                  /* decode.h:38-45 */
  unsigned live;
  uint64_t buff;
  struct in_blk *block;
  const uint32_t *data;
  const uint32_t *limit;
  bool eof;
};
struct retriever_internal_state;
struct decoder_state {              /* decode.h:48-66 */
  struct retriever_internal_state *internal_state;
  bool rand;
  unsigned bwt_idx;
  unsigned block_size;
  uint32_t crc;
  uint32_t ftab[256];
  uint32_t *tt;
  int rle_state;
  uint32_t rle_crc;
  uint32_t rle_index;
  uint32_t rle_avail;
  uint8_t rle_char;
  uint8_t rle_prev;
};
void decoder_init(struct decoder_state *ds);
void decoder_free(struct decoder_state *ds);
int  retrieve(struct decoder_state *ds, struct bitstream *bs);
void decode(struct decoder_state *ds);
int  emit(struct decoder_state *ds, void *buf, size_t *buf_sz);
void lbzamd_decoder_init(struct decoder_state *ds);
void lbzamd_decoder_free(struct decoder_state *ds);
int  lbzamd_retrieve(struct decoder_state *ds, struct bitstream *bs);
void lbzamd_decode(struct decoder_state *ds);
int  lbzamd_emit(struct decoder_state *ds, void *buf, size_t *buf_sz);

/* ---- stage access for parity tests (valid for the last chunk of the last call) ---- */
typedef struct lbzamd_block_info {
  uint32_t n, crc, consumed, bwt_idx, periodic, nmtf, alpha, num_trees, num_sel, out_len, err, rounds;
  uint32_t sort_elems, ticks[8], fticks[16];   /* diagnostics of the BWT stage (batch kernel, deep-tie kernel) */
  uint8_t inuse[256];
} lbzamd_block_info;
enum { LBZAMD_STAGE_RLE = 0, LBZAMD_STAGE_BWT = 1, LBZAMD_STAGE_MTFV = 2, LBZAMD_STAGE_OUT = 3 };
uint32_t lbzamd_block_slots(lbzamd_ctx *ctx);           /* 2 * slabs of the last chunk */
int  lbzamd_block_info_get(lbzamd_ctx *ctx, uint32_t blk, lbzamd_block_info *info);
/* Copies a stage's bytes of block blk to host dst (cap bytes); returns bytes copied or <0.
 * RLE is only valid before the MTF stage ran: use lbzamd_run_stages() to stop early.    */
long lbzamd_read_stage(lbzamd_ctx *ctx, uint32_t blk, int stage, void *dst, size_t cap);
/* Run only stages [0, upto] (0 collect, 1 bwt, 2 mtf, 3 encode) of one chunk of host input. */
int  lbzamd_run_stages(lbzamd_ctx *ctx, const uint8_t *in, size_t len, int upto);

#ifdef __cplusplus
}
#endif
#endif
