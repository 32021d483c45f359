/* lbz_kernels.h -- prototypes of the gfx950 kernels (one .hip file each). */
#ifndef LBZ_KERNELS_H
#define LBZ_KERNELS_H

#include "lbz_dev.h"

struct lbz_stream_state {
  u64 pos;          /* bytes of stream written so far */
  u32 crc;          /* combined CRC so far */
  u32 nblocks;
  u64 n_rle;
  u64 n_mtf;
  u64 sort_elems;
  u32 nperiodic;
  u32 err;
};

/* one compressed block of a stream being decoded (k_decode.hip) */
struct lbz_dblock {
  u64 bit_start;      /* first bit after the block's 48-bit magic */
  u64 bit_used;       /* bit position behind the block's last code (diagnostic) */
  u64 out_off;        /* where the decoded bytes go */
  u32 max_block;      /* bs100k * 100000 of the stream the block belongs to */
  u32 stored_crc, computed_crc;
  u32 randomised, orig_ptr;
  u32 nblock;         /* length before inverse RLE1 */
  u32 out_len;        /* decoded bytes */
  u32 err;            /* 0 ok; 1..9 malformed block; 11 CRC mismatch */
  u32 wk[4];          /* ticks of the walk's parts: sublist lengths, ranking, bytes, inverse RLE1 maps + CRC (diagnostic) */
  u32 cyc;            /* shader clock cycles of the bit chain (diagnostic: with tk[3] the clock the wave ran at) */
  u32 tk[6];          /* 100 MHz ticks of the block's three stages (codes, sort, walk); of the codes stage: bit chain, move-to-front chunks, scan + expansion */
};
#define LBZ_DSCAN_GRID(nbytes) ((u32)((((nbytes) + 14u) / 8u + 255u) / 256u))      /* one thread per aligned 8-byte word */
__global__ void k_dscan(const u8 *in, u64 nbytes, u64 *marks, u32 *nmarks, u32 cap);
__global__ void k_dblock(const u8 *in, u64 nbytes, lbz_dblock *blocks, u32 nblk, u8 *tt8_base, u32 *tt_base, u8 *W_base, u32 *pinfo_base, u8 *X_base, u32 cap);
__global__ void k_dblock_m(const u8 *in, u64 nbytes, lbz_dblock *blocks, u32 nblk, u8 *tt8_base, u32 *tt_base, u8 *W_base, u32 *pinfo_base, u8 *X_base, u32 cap);   /* 512 threads per block */
__global__ void k_dblock_w(const u8 *in, u64 nbytes, lbz_dblock *blocks, u32 nblk, u8 *tt8_base, u32 *tt_base, u8 *W_base, u32 *pinfo_base, u8 *X_base, u32 cap);   /* 1024 threads per block */
__global__ void k_demit(const lbz_dblock *blocks, u32 nblk, const u8 *W_base, const u32 *pinfo_base, u8 *out, u64 out_cap, u32 cap);

/* slabs [first, first + gridDim.x) of the chunk that starts at `in` */
__global__ void k_collect(const u8 *in, u64 in_len, lbz_layout L, u8 *Tbase, lbz_block_meta *meta, u32 first,
                          const u32 *slabs, const u32 *slab_len);
/* Per-round kernels.  A round = slabs [first, first + count), grid = 2 * count: workgroup
 * i < count owns the primary block of slab first + i, workgroup count + i its (usually empty)
 * spill block -- lbz_round_block().  Primaries come first so that the heavy blocks spread over
 * all XCDs (their block ids are all even).  The BWT kernels give workgroup i the workspace slot
 * i of `ws` (count full-size slots, then count spill-size slots at ws_spill).               */
struct lbz_seq_out {
  u64 next;          /* input position behind the last block of this launch */
  u32 nblocks;       /* blocks that took input */
  u32 err;
  u32 nfast;         /* blocks whose cut was found through the step tables */
};
__global__ void k_collect_seq(const u8 *in, u64 in_len, lbz_layout L, u8 *Tbase, lbz_block_meta *meta, u32 nblk,
                              unsigned long long *starts, u32 *ticket, lbz_seq_out *so, u32 slot0,
                              const unsigned long long *carry, const unsigned long long *gpre, u32 ntiles);
/* tables of the sequential mode (k_collect.hip): per 32 KB step of the input */
#define LBZ_SEQ_STEP (LBZ_COLLECT_WG * 64u)
__global__ void k_seq_tiles(const u8 *in, u64 in_len, u32 *last_head, u32 *first_head, u32 *emits);
__global__ void k_seq_prefix(const u32 *last_head, const u32 *first_head, const u32 *emits, u32 ntiles, u32 a0,
                             unsigned long long *carry, unsigned long long *gpre);
/* the partition: a block's rows dealt over `parts` workgroups (ranges of a pass's input), one launch per pass; grid =
 * lbz_seg_grid(nblk, parts).  k_bwt_segs (one workgroup per block) fixes the segments for the kernels behind it. */
__global__ void k_bwt_hist(const u8 *Tbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 parts,
                           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs);
__global__ void k_bwt_scat(const u8 *Tbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 parts,
                           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 pass);
/* ... and the same partition as one workgroup per block, every pass in one launch: rounds that overlap on several streams */
__global__ void k_bwt_part(const u8 *Tbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count,
                           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 segs);
__global__ void k_bwt_segs(lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 segs,
                           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs);
/* from here on a block's sorted rows are dealt over LBZ_BWT_SEGS segment workgroups (k_bwt.hip, "segments"):
 * grid = lbz_seg_grid(nblk), nblk = blocks of the round (2 * count, or count when only primaries are listed) */
__global__ void k_bwt_batch(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first,
                            u32 count, u32 nblk, u32 segs, u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs);
/* the text rounds: launch r of LBZ_DEEP_ROUNDS orders the short runs of tied rows k_bwt_batch listed, strip by strip */
__global__ void k_bwt_deep(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first,
                           u32 count, u32 nblk, u32 segs, u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 round, u32 handover);
__global__ void k_bwt_deepr(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first,
                           u32 count, u32 nblk, u32 segs, u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 round, u32 handover);
/* lbz_block_meta.periodic while the sorter runs: ties left for the rank rounds -- flagged before the third text launch (their
   chain of launches starts there, beside the later text launches), or by the last one (a second chain behind both) */
#ifndef LBZ_DEEP_BUILD
#define LBZ_DEEP_BUILD 1u        /* the text launch whose leftovers get rank entries: the launches behind it (k_bwt_deepr) may step by ranks */
#endif
#define LBZ_DEEP_HANDOVER 1u     /* the text launch that hands over a block with too many rows still tied */
#define LBZ_HANDOVER0 850u       /* thousandths of a block's rows tied as the text rounds begin: above, the block skips them (0: no such rule).  Text-like
                                    blocks of real sources: 60-81 %, the blocks the first text launch used to hand over: 80-99 % (profiles/r05_rows_*.txt) */
#define LBZ_HANDOVER1 400u       /* ... still tied after the first text launch: above, the block goes to the rank rounds */
#define LBZ_TIES_EARLY 2u
#define LBZ_TIES_LATE 3u
/* does a launch of the rank rounds for `which` (LBZ_TIES_EARLY, LBZ_TIES_LATE, or 0: every block with ties left) take this block? */
static inline __host__ __device__ bool lbz_ties_for(u32 flag, u32 which) { return which ? flag == which : flag >= LBZ_TIES_EARLY; }
__global__ void k_bwt_fix0(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first,
                           u32 count, u32 nblk, u32 segs, u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 which);
__global__ void k_bwt_fixr(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first,
                           u32 count, u32 nblk, u32 segs, u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 round, u32 which);
__global__ void k_bwt_fixend(u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count,
                             u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 which);
static inline u32 lbz_seg_grid(u32 nblk, u32 segs) { return (nblk + 7u) / 8u * 8u * segs; }
/* deep-tie rounds a block of up to M rows can need: depths 8 << r < M (keys hold at least 8 symbols) */
static inline u32 lbz_fix_rounds(u32 M) { u32 r = 0; while ((8ull << r) < M) r++; return r; }
__global__ void k_mtf(const u8 *Bbase, u8 *Rbase, u16 *Vbase, u32 *freq_out, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs);
/* ... in two launches for rounds of few blocks: the ranks with `parts` workgroups per block, then zero runs + histogram */
__global__ void k_mtf_ranks(const u8 *Bbase, u8 *Rbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs, u32 parts);
__global__ void k_mtf_zrle(const u8 *Rbase, u16 *Vbase, u32 *freq_out, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs);
__global__ void k_encode(const u16 *Vbase, const u32 *freq_in, u8 *Obase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs);
__global__ void k_offsets(const lbz_block_meta *meta, u32 nblk, u32 bs100k, u32 first, u32 last, u32 body,
                          u64 *offs, lbz_stream_state *st, u8 *out, u64 out_cap);
__global__ void k_gather(const u8 *Obase, const lbz_block_meta *meta, lbz_layout L, const u64 *offs,
                         const lbz_stream_state *st, u8 *out, u32 nslabs);

__global__ void k_meta_pick(const lbz_block_meta *meta, const u32 *slabs, u32 count, u32 *out);
__global__ void k_pool_out(const u8 *Obase, const lbz_block_meta *meta, lbz_layout L, const u32 *slabs, u8 *h_out, u32 *h_pick, u32 parts, u32 spare);
__global__ void k_pool_split(const lbz_block_meta *meta, const u32 *slabs, u32 count, u32 *fast, u32 *slow, u32 spare);
__global__ void k_pool_in(const u8 *h_in, u8 *d_in, u32 M, const u32 *slabs, const u32 *lens, u32 parts);

__device__ __forceinline__ u32 lbz_queue_block(u32 q, u32 nslabs)      /* whole chunk: k_gather */
{
  return q < nslabs ? 2u * q : 2u * (q - nslabs) + 1u;
}
__device__ __forceinline__ u32 lbz_round_block(u32 first, u32 count, u32 i, const u32 *slabs = nullptr)
{
  /* slabs: the round's slabs listed one by one (the work-unit interface batches whatever slabs
     its callers hold); nullptr: slabs first .. first + count - 1 */
  const u32 k = i < count ? i : i - count;
  return 2u * (slabs ? slabs[k] : first + k) + (i < count ? 0u : 1u);
}

#endif
