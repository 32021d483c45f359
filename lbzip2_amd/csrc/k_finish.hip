/*
 * k_finish.hip -- stream assembly on the device: block offsets, the stream CRC fold and the
 * gather of the byte-aligned blocks into one contiguous .bz2 stream.
 *
 * Replaces, for the batch path, do_reorder()/write_header()/write_trailer() of the
 * reference's src/compress.c:238-250, 291-321 and the combine_crc macro (encode.h:38):
 * the stream is  "BZh"+level | blocks in slab order | 0x177245385090 | combined CRC.
 * Every block is a whole number of bytes (encode.c:514-525), so assembly is a prefix sum of
 * block sizes plus a copy.  The fold  cc' = rotl(cc,1) ^ ~crc  is order dependent and tiny:
 * one lane walks the blocks.
 */
#include "lbz_common.h"
#undef LBZ_WG
#define LBZ_WG LBZ_FINISH_WG
#undef LBZ_NW
#define LBZ_NW (LBZ_WG / 64)
#include "lbz_kernels.h"


/* one workgroup; offs[b] = absolute stream offset of block b of this chunk.  Sizes are
 * prefix-summed by the whole workgroup (tiles of LBZ_WG blocks); the CRC fold, which is order
 * dependent (rotate, then xor), runs on one lane over values staged in LDS.               */
__global__ void __launch_bounds__(LBZ_WG)
k_offsets(const lbz_block_meta *meta, u32 nblk, u32 bs100k, u32 first, u32 last, u32 body,
          u64 *offs, lbz_stream_state *st, u8 *out, u64 out_cap)
{
  __shared__ wg_scratch sc;
  __shared__ u32 crcs[LBZ_WG];
  __shared__ u32 live[LBZ_WG];
  __shared__ u32 fold;
  const u32 tid = threadIdx.x;
  u64 pos = first ? (body ? 0ull : 4ull) : st->pos;     /* body: blocks only, no header/trailer (multi-GPU shards) */
  u32 nb = 0, nper = 0, err = 0;
  u64 nrle = 0, nmtf = 0, nsort = 0;
  if (tid == 0) fold = first ? 0u : st->crc;
  __syncthreads();
  for (u32 t0 = 0; t0 < nblk; t0 += LBZ_WG) {
    const u32 b = t0 + tid;
    u32 len = 0, on = 0;
    if (b < nblk && meta[b].n != 0u) {
      const lbz_block_meta *m = &meta[b];
      on = 1; len = m->out_len;
      crcs[tid] = m->crc;
      if (m->err) err = m->err;
      nb++; nrle += m->n; nmtf += m->nmtf; nsort += m->sort_elems; nper += m->periodic;
    }
    live[tid] = on;
    u32 tot;
    const u32 ex = wg_excl_add(len, &tot, &sc);
    if (b < nblk) offs[b] = pos + ex;
    if (tid == 0) {
      u32 cc = fold;
      const u32 cnt = nblk - t0 < LBZ_WG ? nblk - t0 : LBZ_WG;
      for (u32 i = 0; i < cnt; i++) if (live[i]) cc = ((cc << 1) | (cc >> 31)) ^ ~crcs[i];
      fold = cc;
    }
    pos += tot;
    __syncthreads();
  }
  /* statistics: per-thread partial sums -> lane 0 */
  const u32 tnb = wg_sum(nb, &sc), tper = wg_sum(nper, &sc), terr = wg_max(err, &sc);
  const u32 rle_lo = wg_sum((u32)(nrle & 0xFFFFFu), &sc), rle_hi = wg_sum((u32)(nrle >> 20), &sc);
  const u32 mtf_lo = wg_sum((u32)(nmtf & 0xFFFFFu), &sc), mtf_hi = wg_sum((u32)(nmtf >> 20), &sc);
  const u32 srt_lo = wg_sum((u32)(nsort & 0xFFFFFu), &sc), srt_hi = wg_sum((u32)(nsort >> 20), &sc);
  if (tid != 0) return;
  if (first) {
    st->nblocks = 0; st->n_rle = 0; st->n_mtf = 0; st->sort_elems = 0; st->nperiodic = 0; st->err = 0;
    if (!body && out_cap >= 4) { out[0] = 'B'; out[1] = 'Z'; out[2] = 'h'; out[3] = (u8)('0' + bs100k); }
  }
  st->nblocks += tnb; st->nperiodic += tper;
  st->n_rle += ((u64)rle_hi << 20) + rle_lo;
  st->n_mtf += ((u64)mtf_hi << 20) + mtf_lo;
  st->sort_elems += ((u64)srt_hi << 20) + srt_lo;
  if (terr) st->err = terr;
  const u32 cc = fold;
  if (pos + ((last && !body) ? 10u : 0u) > out_cap) { st->err = 100u; st->pos = pos; st->crc = cc; return; }
  if (last && !body) {
    const u8 tr[6] = { 0x17, 0x72, 0x45, 0x38, 0x50, 0x90 };
    for (u32 i = 0; i < 6; i++) out[pos + i] = tr[i];
    out[pos + 6] = (u8)(cc >> 24); out[pos + 7] = (u8)(cc >> 16);
    out[pos + 8] = (u8)(cc >> 8);  out[pos + 9] = (u8)cc;
    pos += 10;
  }
  st->pos = pos;
  st->crc = cc;
}

/* byte copy of the blocks into their (arbitrarily aligned) stream positions: a workgroup per block when `out` is device
 * memory; when `out` is page-locked host memory the copy runs at the speed of the link, so a few small workgroups walk
 * the blocks (grid-stride) -- sixteen-wave workgroups waiting on PCIe writes would hold the CUs the other rounds need */
__global__ void __launch_bounds__(LBZ_WG)
k_gather(const u8 *Obase, const lbz_block_meta *meta, lbz_layout L, const u64 *offs,
         const lbz_stream_state *st, u8 *out, u32 nslabs)
{
  if (st->err) return;
  for (u32 q = blockIdx.x; q < 2u * nslabs; q += gridDim.x) {
    const u32 blk = lbz_queue_block(q, nslabs);
    const lbz_block_meta *m = &meta[blk];
    if (m->n == 0u) continue;
    const u8 *src = Obase + lbz_out_off(L, blk);
    u8 *dst = out + offs[blk];
    const u32 len = m->out_len;
    /* dword stores where dst is aligned; src is always 4-byte aligned */
    const u32 head = (u32)((4u - ((uintptr_t)dst & 3u)) & 3u);
    const u32 h = head < len ? head : len;
    if (threadIdx.x < h) dst[threadIdx.x] = src[threadIdx.x];
    const u32 nw = (len - h) >> 2;
    u32 *d32 = reinterpret_cast<u32 *>(dst + h);
    for (u32 i = threadIdx.x; i < nw; i += blockDim.x) {
      const u8 *s = src + h + 4u * i;
      d32[i] = (u32)s[0] | ((u32)s[1] << 8) | ((u32)s[2] << 16) | ((u32)s[3] << 24);
    }
    const u32 done = h + 4u * nw;
    if (threadIdx.x < len - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
  }
}

/* work-unit rounds: the four words a caller waits for, of the listed slabs' primary blocks, packed for one
 * small device-to-host copy (instead of the records of the whole pool) */
__global__ void __launch_bounds__(256)
k_meta_pick(const lbz_block_meta *meta, const u32 *slabs, u32 count, u32 *out)
{
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  const lbz_block_meta *m = &meta[2u * slabs[i]];
  out[4u * i] = m->consumed; out[4u * i + 1u] = m->out_len; out[4u * i + 2u] = m->crc; out[4u * i + 3u] = m->err;
}

/* work-unit encode rounds: the packed blocks of the listed slabs go to the pool's page-locked staging area -- the device
 * writes them there itself (whole words, 16 bytes a thread; PCIe writes are posted), their four result words beside them, so
 * that a round ends with ONE wait of its leader instead of a copy of the sizes, a wait, a copy per block and another wait.
 * grid = parts * count.                                                                                               */
__global__ void __launch_bounds__(256)
k_pool_out(const u8 *Obase, const lbz_block_meta *meta, lbz_layout L, const u32 *slabs, u8 *h_out, u32 *h_pick, u32 parts, u32 spare)
{
  const u32 i = blockIdx.x / parts, part = blockIdx.x % parts, slab = slabs[i];
  if (slab == spare) return;                               /* (an entry of the other half of a split list) */
  const lbz_block_meta *m = &meta[2u * slab];
  if (part == 0u && threadIdx.x == 0u) {
    h_pick[4u * i] = m->consumed; h_pick[4u * i + 1u] = m->out_len; h_pick[4u * i + 2u] = m->crc; h_pick[4u * i + 3u] = m->err;
  }
  if (m->err) return;
  const u32 words16 = (m->out_len + 15u) / 16u;            /* (the block's reserve is a multiple of 256 bytes) */
  const uint4 *src = reinterpret_cast<const uint4 *>(Obase + lbz_out_off(L, 2u * slab));
  uint4 *dst = reinterpret_cast<uint4 *>(h_out + (size_t)slab * L.out_a);
  for (u32 w = part * 256u + threadIdx.x; w < words16; w += parts * 256u) dst[w] = src[w];
}

/* work-unit collect rounds: the callers' slabs come from the page-locked staging area by the device's own reads (16 bytes a
 * thread, every workgroup with four loads in flight per thread) -- one launch instead of a copy call per slab. grid = parts * count */
__global__ void __launch_bounds__(256)
k_pool_in(const u8 *h_in, u8 *d_in, u32 M, const u32 *slabs, const u32 *lens, u32 parts)
{
  const u32 i = blockIdx.x / parts, part = blockIdx.x % parts, slab = slabs[i];
  const u32 words16 = (lens[i] + 15u) / 16u;
  const uint4 *src = reinterpret_cast<const uint4 *>(h_in + (size_t)slab * M);
  uint4 *dst = reinterpret_cast<uint4 *>(d_in + (size_t)slab * M);
  const u32 stride = parts * 256u;
  u32 w = part * 256u + threadIdx.x;
  for (; w + 3u * stride < words16; w += 4u * stride) {
    const uint4 a = src[w], b = src[w + stride], c = src[w + 2u * stride], d = src[w + 3u * stride];
    dst[w] = a; dst[w + stride] = b; dst[w + 2u * stride] = c; dst[w + 3u * stride] = d;
  }
  for (; w < words16; w += stride) dst[w] = src[w];
}

/* work-unit encode rounds: the round's list in two of the same length -- the blocks the text rounds finished, and the ones
 * the rank rounds still have to sort (flagged LBZ_TIES_*); where a list lacks a block it names the pool's spare slab. */
__global__ void __launch_bounds__(256)
k_pool_split(const lbz_block_meta *meta, const u32 *slabs, u32 count, u32 *fast, u32 *slow, u32 spare)
{
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  const u32 slab = slabs[i];
  const bool left = meta[2u * slab].n >= 2u && meta[2u * slab].periodic >= LBZ_TIES_EARLY;
  fast[i] = left ? spare : slab;
  slow[i] = left ? slab : spare;
}
