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

/* slabs [first, first + gridDim.x) of the chunk that starts at `in` */
__global__ void k_collect(const u8 *in, u64 in_len, lbz_layout L, u8 *Tbase, lbz_block_meta *meta, u32 first,
                          const u32 *slabs, const u32 *slab_len);
/* Per-round kernels.  A round = slabs [first, first + count), grid = 2 * count: workgroup
 * i < count owns the primary block of slab first + i, workgroup count + i its (usually empty)
 * spill block -- lbz_round_block().  Primaries come first so that the heavy blocks spread over
 * all XCDs (their block ids are all even).  The BWT kernels give workgroup i the workspace slot
 * i of `ws` (count full-size slots, then count spill-size slots at ws_spill).               */
__global__ void k_bwt_part(const u8 *Tbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count,
                           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs);
__global__ void k_bwt_part2(const u8 *Tbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count,
                            u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs);
__global__ void k_bwt_batch(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first,
                            u32 count, u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs);
__global__ void k_bwt_fix(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first,
                          u32 count, u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs);
__global__ void k_mtf(const u8 *Bbase, u8 *Rbase, u16 *Vbase, u32 *freq_out, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs);
__global__ void k_encode(const u16 *Vbase, const u32 *freq_in, u8 *Obase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs);
__global__ void k_offsets(const lbz_block_meta *meta, u32 nblk, u32 bs100k, u32 first, u32 last, u32 body,
                          u64 *offs, lbz_stream_state *st, u8 *out, u64 out_cap);
__global__ void k_gather(const u8 *Obase, const lbz_block_meta *meta, lbz_layout L, const u64 *offs,
                         const lbz_stream_state *st, u8 *out, u32 nslabs);

__global__ void k_meta_pick(const lbz_block_meta *meta, const u32 *slabs, u32 count, u32 *out);

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
