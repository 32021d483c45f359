/*
 * k_decode.hip -- the inverse path: block-parallel bzip2 decoding,
 * one compressed block per workgroup, every block of a stream in flight at once.
 *
 * Follows the stages of the reference's decompressor -- scan() for block magics (src/parse.c:282),
 * retrieve() = prefix-code decoding + inverse MTF / zero-run expansion (src/decode.c:519-850),
 * decode() = the counting sort that turns the BWT string into a linked list (src/decode.c:852-942),
 * emit() = the list walk + inverse RLE1 + CRC (src/decode.c:944-1146) -- but the unit of parallelism
 * is the block, not the thread: three of the four stages are serial chains by nature (a bit cursor,
 * a move-to-front list, a pointer chase through a 3.6 MB array), so such a stage runs on ONE lane per
 * block and a thousand blocks hide each other's latency; the counting sort is wide.  The pointer
 * chase is bound by HBM latency (one dependent 4-byte load per output byte), not by bandwidth.
 *
 *   k_dscan   every bit position of the stream is tested for the two 48-bit magics
 *   k_dhuff   header, code tables, prefix-code decoding, inverse MTF, RUNA/RUNB expansion -> BWT bytes
 *   k_dsort   cftab + stable counting sort -> tt[] (next pointer << 8 | byte)
 *   k_dwalk   the walk: RLE1'd bytes W[], decoded size and CRC (inverse RLE1 state machine in the loop)
 *   k_demit   inverse RLE1 of W[] into the output at the block's offset
 */
#include "lbz_kernels.h"

#define DEC_MAX_SEL 18002u

struct dec_lds {
  int limit[LBZ_MAX_TREES][24];       /* per code length l: largest 20-bit window whose top l bits are a code of length <= l (-1: none) */
  int base[LBZ_MAX_TREES][24];        /* perm index = (window >> (20 - l)) - base */
  u16 perm[LBZ_MAX_TREES][LBZ_MAX_ALPHA + 2];
  u8 minlen[LBZ_MAX_TREES], maxlen[LBZ_MAX_TREES];
  u8 len[LBZ_MAX_TREES][LBZ_MAX_ALPHA + 2];
  u8 seq2unseq[256];
  u8 mtf[256];
  u32 unzftab[256];
};

/* MSB-first bit cursor over the stream in global memory (one lane) */
struct bitrd {
  const u8 *in;
  u64 nbytes;
  u64 pos;        /* next byte to load */
  u64 buf;        /* left-aligned */
  u32 live;
};
__device__ __forceinline__ void br_fill(bitrd *b)
{
  while (b->live <= 56u) {
    const u64 v = b->pos < b->nbytes ? b->in[b->pos] : 0u;
    b->pos++;
    b->buf |= v << (56u - b->live);
    b->live += 8u;
  }
}
__device__ __forceinline__ void br_init(bitrd *b, const u8 *in, u64 nbytes, u64 bitpos)
{
  b->in = in; b->nbytes = nbytes; b->pos = bitpos >> 3; b->buf = 0; b->live = 0;
  br_fill(b);
  const u32 skip = (u32)(bitpos & 7u);
  b->buf <<= skip; b->live -= skip;
}
__device__ __forceinline__ u32 br_get(bitrd *b, u32 n)      /* 1 <= n <= 32 */
{
  if (b->live < n) br_fill(b);
  const u32 v = (u32)(b->buf >> (64u - n));
  b->buf <<= n; b->live -= n;
  return v;
}
__device__ __forceinline__ u32 br_peek20(bitrd *b)
{
  if (b->live < 20u) br_fill(b);
  return (u32)(b->buf >> 44);
}
__device__ __forceinline__ void br_skip(bitrd *b, u32 n) { b->buf <<= n; b->live -= n; }
__device__ __forceinline__ u64 br_bitpos(const bitrd *b) { return b->pos * 8ull - b->live; }

/* ------------------------------------------------------------------ k_dscan */
/* marks[]: bit position << 1 | kind (0 = block magic 0x314159265359, 1 = end-of-stream magic
 * 0x177245385090), in no particular order (the host sorts the handful of them). */
__global__ void __launch_bounds__(256)
k_dscan(const u8 *in, u64 nbytes, u64 *marks, u32 *nmarks, u32 cap)
{
  const u64 p = (u64)blockIdx.x * 256u + threadIdx.x;
  if (p + 6u > nbytes) return;
  u64 v = 0;
  for (u32 i = 0; i < 8u; i++) v = (v << 8) | (u64)(p + i < nbytes ? in[p + i] : 0u);
  for (u32 s = 0; s < 8u; s++) {
    if (s && p + 7u > nbytes) break;
    const u64 w = (v >> (16u - s)) & 0xFFFFFFFFFFFFull;
    const int kind = w == 0x314159265359ull ? 0 : (w == 0x177245385090ull ? 1 : -1);
    if (kind >= 0) {
      const u32 k = atomicAdd(nmarks, 1u);
      if (k < cap) marks[k] = ((p * 8ull + s) << 1) | (u64)kind;
    }
  }
}

/* ------------------------------------------------------------------ k_dhuff */
/* One wave per block, lane 0 works: the whole stage is one dependent chain (bit cursor -> code ->
 * move-to-front list -> output position).                                                        */
__global__ void __launch_bounds__(64)
k_dhuff(const u8 *in, u64 nbytes, lbz_dblock *blocks, u32 nblk, u8 *tt8_base, u32 *ftab_base, u8 *sel_base, u32 cap)
{
  __shared__ dec_lds S;
  const u32 lane = threadIdx.x;
  const u32 blk = blockIdx.x;
  if (blk >= nblk) return;
  lbz_dblock *D = &blocks[blk];
  u8 *tt8 = tt8_base + (size_t)blk * cap;
  u8 *sel = sel_base + (size_t)blk * DEC_MAX_SEL;
  const u32 maxn = D->max_block < cap ? D->max_block : cap;
  for (u32 i = lane; i < 256u; i += 64u) { S.unzftab[i] = 0; S.mtf[i] = (u8)i; }
  __syncthreads();
  if (lane == 0u) {
    bitrd b;
    br_init(&b, in, nbytes, D->bit_start);
    u32 err = 0, n = 0;
    D->stored_crc = br_get(&b, 32);
    D->randomised = br_get(&b, 1);
    D->orig_ptr = br_get(&b, 24);
    /* used-byte map */
    const u32 big = br_get(&b, 16);
    u32 ninuse = 0;
    for (u32 i = 0; i < 16u; i++)
      if (big & (0x8000u >> i)) {
        const u32 small = br_get(&b, 16);
        for (u32 j = 0; j < 16u; j++) if (small & (0x8000u >> j)) S.seq2unseq[ninuse++] = (u8)(16u * i + j);
      }
    if (ninuse == 0u) err = 1;
    const u32 alpha = ninuse + 2u, eob = alpha - 1u;
    const u32 ngroups = br_get(&b, 3);
    const u32 nsel = br_get(&b, 15);
    if (!err && (ngroups < 2u || ngroups > LBZ_MAX_TREES || nsel < 1u)) err = 2;
    /* selectors: unary move-to-front codes */
    if (!err) {
      u8 order[LBZ_MAX_TREES];
      for (u32 i = 0; i < LBZ_MAX_TREES; i++) order[i] = (u8)i;
      for (u32 i = 0; i < nsel && !err; i++) {
        u32 j = 0;
        while (br_get(&b, 1)) { j++; if (j >= ngroups) { err = 3; break; } }
        if (err) break;
        const u8 t = order[j];
        for (u32 k = j; k > 0; k--) order[k] = order[k - 1u];
        order[0] = t;
        if (i < DEC_MAX_SEL) sel[i] = t;
      }
    }
    /* code lengths: 5-bit start, then +1 / -1 steps (what encode.c:1231-1255 writes) */
    for (u32 t = 0; t < ngroups && !err; t++) {
      int cur = (int)br_get(&b, 5);
      for (u32 v = 0; v < alpha && !err; v++) {
        for (;;) {
          if (cur < 1 || cur > 20) { err = 4; break; }
          if (!br_get(&b, 1)) break;
          cur += br_get(&b, 1) ? -1 : 1;
        }
        S.len[t][v] = (u8)cur;
      }
    }
    /* canonical decoding tables (codes of one length are consecutive, lengths ascend; the format's own rule) */
    for (u32 t = 0; t < ngroups && !err; t++) {
      u32 mn = 32, mx = 0;
      for (u32 v = 0; v < alpha; v++) { const u32 l = S.len[t][v]; mn = l < mn ? l : mn; mx = l > mx ? l : mx; }
      S.minlen[t] = (u8)mn; S.maxlen[t] = (u8)mx;
      u32 pp = 0, vec = 0;
      for (u32 l = mn; l <= mx; l++) {
        u32 cnt = 0;
        const u32 before = pp;                                    /* symbols with a shorter code */
        for (u32 v = 0; v < alpha; v++) if (S.len[t][v] == l) { S.perm[t][pp++] = (u16)v; cnt++; }
        const u32 first = vec;                                    /* first code of this length */
        vec += cnt;
        S.limit[t][l] = (int)(vec << (20u - l)) - 1;              /* vec - 1 is the last code of length <= l; -1 if none */
        S.base[t][l] = (int)first - (int)before;
        vec <<= 1;
      }
    }
    /* symbols */
    if (!err) {
      u32 groupno = 0, grouppos = 0, t = 0;
      u32 es = 0, N = 1;
      for (;;) {
        if (grouppos == 0u) {
          if (groupno >= nsel) { err = 5; break; }
          t = sel[groupno++]; grouppos = LBZ_GROUP;
        }
        grouppos--;
        const u32 code = br_peek20(&b);
        u32 l = S.minlen[t];
        const u32 mx = S.maxlen[t];
        while (l <= mx && (int)code > S.limit[t][l]) l++;
        if (l > mx) { err = 6; break; }
        br_skip(&b, l);
        const int pi = (int)(code >> (20u - l)) - S.base[t][l];
        if (pi < 0 || pi >= (int)alpha) { err = 6; break; }
        const u32 sym = S.perm[t][pi];
        if (sym <= 1u) {                                         /* RUNA / RUNB: bijective base-2 digits of a zero run */
          es += (sym == 0u ? N : 2u * N);
          N <<= 1;
          if (N > (1u << 21)) { err = 7; break; }
          continue;
        }
        if (es) {
          const u8 uc = S.seq2unseq[S.mtf[0]];
          if (n + es > maxn) { err = 8; break; }
          for (u32 i = 0; i < es; i++) tt8[n + i] = uc;
          S.unzftab[uc] += es;
          n += es; es = 0; N = 1;
        }
        if (sym == eob) break;
        const u32 nn = sym - 1u;
        const u8 m = S.mtf[nn];
        for (u32 k = nn; k > 0; k--) S.mtf[k] = S.mtf[k - 1u];
        S.mtf[0] = m;
        const u8 uc = S.seq2unseq[m];
        if (n >= maxn) { err = 8; break; }
        tt8[n++] = uc;
        S.unzftab[uc]++;
      }
    }
    if (!err && D->randomised) err = 10;                         /* obsolete format variant, never written by lbzip2 */
    if (!err && (n == 0u || D->orig_ptr >= n)) err = 9;
    D->nblock = err ? 0u : n;
    D->err = err;
    D->bit_used = br_bitpos(&b);
  }
  __syncthreads();
  u32 *ftab = ftab_base + (size_t)blk * 256u;
  for (u32 i = lane; i < 256u; i += 64u) ftab[i] = S.unzftab[i];
}

/* ------------------------------------------------------------------ k_dsort */
/* tt[k] = (position of the k-th byte in sorted order) << 8 | byte at position k  (decode.c:852-942).
 * Stable counting sort, 256 positions at a time: ranks inside a wave from match-any ballots, the four
 * waves of a tile in order through per-wave digit counts.                                          */
__global__ void __launch_bounds__(256)
k_dsort(const lbz_dblock *blocks, u32 nblk, const u8 *tt8_base, const u32 *ftab_base, u32 *tt_base, u32 cap)
{
  __shared__ u32 cf[256];
  __shared__ u32 wcnt[4][256];
  __shared__ u32 wsum[4];
  const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const u32 blk = blockIdx.x;
  if (blk >= nblk) return;
  const lbz_dblock *D = &blocks[blk];
  const u32 n = D->nblock;
  if (D->err || n == 0u) return;
  const u8 *tt8 = tt8_base + (size_t)blk * cap;
  u32 *tt = tt_base + (size_t)blk * cap;
  {
    const u32 c = ftab_base[(size_t)blk * 256u + tid];
    u32 inc = c;
    for (u32 d = 1; d < 64u; d <<= 1) { const u32 o = (u32)__shfl_up((int)inc, d); if (lane >= d) inc += o; }
    if (lane == 63u) wsum[w] = inc;
    __syncthreads();
    u32 basev = 0;
    for (u32 i = 0; i < w; i++) basev += wsum[i];
    cf[tid] = basev + inc - c;
  }
  for (u32 i = tid; i < n; i += 256u) tt[i] = tt8[i];
  __syncthreads();
  for (u32 t0 = 0; t0 < n; t0 += 256u) {
    for (u32 i = tid; i < 1024u; i += 256u) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    const u32 i = t0 + tid;
    const bool ok = i < n;
    const u32 d = ok ? tt8[i] : 0u;
    u64 mask = __ballot(ok);
#pragma unroll
    for (u32 bb = 0; bb < 8u; bb++) {
      const bool bit = (d >> bb) & 1u;
      const u64 bal = __ballot(bit);
      mask &= bit ? bal : ~bal;
    }
    const u32 below = (u32)__popcll(mask & ((1ull << lane) - 1ull));
    if (ok && below == 0u) wcnt[w][d] = (u32)__popcll(mask);
    __syncthreads();
    if (ok) {
      u32 dst = cf[d] + below;
      for (u32 w2 = 0; w2 < w; w2++) dst += wcnt[w2][d];
      tt[dst] |= i << 8;
    }
    __syncthreads();
    cf[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
    __syncthreads();
  }
}

/* ------------------------------------------------------------------ k_dwalk */
__device__ __forceinline__ u32 dec_crc_step(const u32 *tab, u32 crc, u32 byte) { return (crc << 8) ^ tab[(crc >> 24) ^ byte]; }

/* One wave per block, lane 0 walks: n dependent loads.  The inverse-RLE1 state machine runs under the load
 * latency: decoded length and CRC-32 (poly 0x04C11DB7, MSB first) come out of the same loop; the RLE1'd bytes
 * are kept (W) so that the output pass needs no second chase.                                          */
__global__ void __launch_bounds__(64)
k_dwalk(lbz_dblock *blocks, u32 nblk, const u32 *tt_base, u8 *W_base, u32 cap)
{
  __shared__ u32 crctab[256];
  const u32 lane = threadIdx.x;
  const u32 blk = blockIdx.x;
  if (blk >= nblk) return;
  for (u32 i = lane; i < 256u; i += 64u) {
    u32 c = i << 24;
    for (u32 k = 0; k < 8u; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : c << 1;
    crctab[i] = c;
  }
  __syncthreads();
  if (lane != 0u) return;
  lbz_dblock *D = &blocks[blk];
  const u32 n = D->nblock;
  if (D->err || n == 0u) { D->out_len = 0; return; }
  const u32 *tt = tt_base + (size_t)blk * cap;
  u8 *W = W_base + (size_t)blk * cap;
  u32 tpos = tt[D->orig_ptr] >> 8;
  u32 crc = 0xFFFFFFFFu, run = 0, prev = 256u;
  u64 outlen = 0;
  for (u32 k = 0; k < n; k++) {
    const u32 x = tt[tpos];
    const u32 ch = x & 255u;
    tpos = x >> 8;
    W[k] = (u8)ch;
    if (run == 4u) {                                            /* a count byte: ch more copies of prev */
      for (u32 r = 0; r < ch; r++) crc = dec_crc_step(crctab, crc, prev);
      outlen += ch;
      run = 0; prev = 256u;
      continue;
    }
    if (ch == prev) run++; else { run = 1; prev = ch; }
    crc = dec_crc_step(crctab, crc, ch);
    outlen++;
  }
  D->computed_crc = ~crc;
  D->out_len = outlen > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)outlen;
  if (D->computed_crc != D->stored_crc) D->err = 11;
}

/* ------------------------------------------------------------------ k_demit */
/* inverse RLE1 of W[] to out + out_off: lane 0 runs the state machine, the wave writes the long runs */
__global__ void __launch_bounds__(64)
k_demit(const lbz_dblock *blocks, u32 nblk, const u8 *W_base, u8 *out, u64 out_cap, u32 cap)
{
  const u32 lane = threadIdx.x;
  const u32 blk = blockIdx.x;
  if (blk >= nblk) return;
  const lbz_dblock *D = &blocks[blk];
  const u32 n = D->nblock;
  if (D->err || n == 0u || lane != 0u) return;
  if (D->out_off + D->out_len > out_cap) return;
  const u8 *W = W_base + (size_t)blk * cap;
  u8 *o = out + D->out_off;
  u32 run = 0, prev = 256u;
  for (u32 k = 0; k < n; k++) {
    const u32 ch = W[k];
    if (run == 4u) {
      for (u32 r = 0; r < ch; r++) *o++ = (u8)prev;
      run = 0; prev = 256u;
      continue;
    }
    if (ch == prev) run++; else { run = 1; prev = ch; }
    *o++ = (u8)ch;
  }
}
