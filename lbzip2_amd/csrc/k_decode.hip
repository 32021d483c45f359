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
/* The walk (decode.c:944-1146 is one pointer chase per block: n dependent loads) done as LIST RANKING so
 * that a block has hundreds of chases in flight instead of one:
 *
 *   1. every 512th list node (and the start node) is a splitter; from each splitter a lane follows the list
 *      to the next splitter and records (steps, splitter reached).  Lanes take splitters from a counter,
 *      so a long sublist does not hold the others up;
 *   2. one lane ranks the <= 1760 splitters: sublist k starts at output offset off[k].  A list that closes
 *      before n steps (the block is periodic: the BWT permutation has several cycles) gives the period;
 *   3. the sublists are followed again, bytes go to W[off[k] + j]; a periodic block is filled from its
 *      first period;
 *   4. inverse RLE1 without the serial state machine: the state (bytes of the current run seen, 0..4; 4 =
 *      "next byte is a count") moves by c -> c+1 mod 5 on a byte equal to its predecessor and by
 *      c -> (c == 4 ? 0 : 1) otherwise, so a chunk of W is a map {0..4} -> {0..4}; the 256 chunk maps are
 *      composed in order, then every lane re-reads its chunk with the right start state: decoded length,
 *      CRC (from 0) and, per 16 bytes of W, the output offset + state that k_demit starts from;
 *   5. CRC-32 is linear: crc(A|B) = crc(A) * x^(8|B|) + crc(B) over GF(2)[x]/P -- the chunk CRCs are
 *      shifted by the decoded length behind them (square-and-multiply with x^(8 * 2^k)) and xor-ed.      */
#define DW_T 256u
#define DW_LOG 9u
#define DW_STRIDE (1u << DW_LOG)
#define DW_MAXS ((LBZ_MAX_BLOCK >> DW_LOG) + 4u)
#define DW_NONE 0xFFFFFFFFu
#define CRC_POLY 0x04C11DB7u

struct walk_lds {
  u32 len[DW_MAXS], nxt[DW_MAXS], off[DW_MAXS];
  u32 crctab[256];
  u32 pow8[32];
  u32 fn[DW_T];          /* chunk maps, 3 bits per start state */
  u32 olen[DW_T];        /* decoded bytes of the chunk, then the exclusive prefix */
  u32 ctr, ctr2, period, total;
  u32 xr[4];
};

__device__ __forceinline__ u32 dec_crc_step(const u32 *tab, u32 crc, u32 byte) { return (crc << 8) ^ tab[(crc >> 24) ^ byte]; }
__device__ __forceinline__ u32 gf2_mulmod(u32 a, u32 b)
{
  u32 r = 0;
  for (int i = 31; i >= 0; i--) {
    r = (r << 1) ^ ((r & 0x80000000u) ? CRC_POLY : 0u);
    if ((b >> i) & 1u) r ^= a;
  }
  return r;
}
__device__ __forceinline__ u32 crc_shift(const u32 *pow8, u32 v, u32 nbytes)     /* v * x^(8 nbytes) mod P */
{
  for (u32 k = 0; nbytes; k++, nbytes >>= 1) if (nbytes & 1u) v = gf2_mulmod(v, pow8[k]);
  return v;
}
__device__ __forceinline__ u32 rle_step(u32 c, bool eq) { return eq ? (c == 4u ? 0u : c + 1u) : (c == 4u ? 0u : 1u); }

__global__ void __launch_bounds__(DW_T)
k_dwalk(lbz_dblock *blocks, u32 nblk, const u32 *tt_base, u8 *W_base, u32 *pinfo_base, u32 cap)
{
  __shared__ walk_lds S;
  const u32 tid = threadIdx.x;
  const u32 blk = blockIdx.x;
  if (blk >= nblk) return;
  lbz_dblock *D = &blocks[blk];
  const u32 n = D->nblock;
  if (D->err || n == 0u) { if (tid == 0u) D->out_len = 0; return; }
  const u32 *tt = tt_base + (size_t)blk * cap;
  u8 *W = W_base + (size_t)blk * cap;
  u32 *pinfo = pinfo_base + (size_t)blk * (cap / 16u);

  {
    u32 c = tid << 24;
    for (u32 k = 0; k < 8u; k++) c = (c & 0x80000000u) ? (c << 1) ^ CRC_POLY : c << 1;
    S.crctab[tid] = c;
  }
  if (tid == 0u) {
    u32 p = 0x100u;                                   /* x^8 */
    for (u32 k = 0; k < 32u; k++) { S.pow8[k] = p; p = gf2_mulmod(p, p); }
    S.ctr = 0; S.ctr2 = 0;
  }
  const u32 t0 = tt[D->orig_ptr] >> 8;
  const u32 ns0 = (n + DW_STRIDE - 1u) >> DW_LOG;    /* splitters k * 512 < n */
  const bool extra = (t0 & (DW_STRIDE - 1u)) != 0u;
  const u32 ns = ns0 + (extra ? 1u : 0u);
  const u32 start_id = extra ? ns0 : t0 >> DW_LOG;
  for (u32 i = tid; i < ns; i += DW_T) S.off[i] = DW_NONE;
  __syncthreads();

  /* 1. sublist lengths */
  {
    u32 k = atomicAdd(&S.ctr, 1u);
    u32 node = k < ns0 ? k << DW_LOG : t0, cnt = 0;
    while (k < ns) {
      node = tt[node] >> 8;
      cnt++;
      if ((node & (DW_STRIDE - 1u)) == 0u || node == t0 || cnt >= n) {
        S.len[k] = cnt;
        S.nxt[k] = node == t0 ? start_id : node >> DW_LOG;
        k = atomicAdd(&S.ctr, 1u);
        node = k < ns0 ? k << DW_LOG : t0;
        cnt = 0;
      }
    }
  }
  __syncthreads();
  /* 2. rank the splitters */
  if (tid == 0u) {
    u32 k = start_id, total = 0, steps = 0;
    do {
      S.off[k] = total;
      total += S.len[k];
      k = S.nxt[k];
    } while (k != start_id && total < n && ++steps < ns);
    S.period = total < n ? total : n;
  }
  __syncthreads();
  /* 3. bytes */
  {
    u32 k = DW_NONE, node = 0, o = 0, left = 0;
    for (;;) {
      if (left == 0u) {
        k = atomicAdd(&S.ctr2, 1u);
        if (k >= ns) break;
        o = S.off[k];
        if (o == DW_NONE) continue;
        left = S.len[k];
        if (o + left > n) left = n - o;
        node = k < ns0 ? k << DW_LOG : t0;
        if (left == 0u) continue;
      }
      const u32 x = tt[node];
      W[o++] = (u8)x;
      node = x >> 8;
      left--;
    }
  }
  __syncthreads();
  const u32 period = S.period;
  if (period < n) {
    for (u32 j = period + tid; j < n; j += DW_T) W[j] = W[j % period];
    __syncthreads();
  }

  /* 4. chunk maps */
  const u32 npiece = (n + 15u) / 16u;
  const u32 ppc = (npiece + DW_T - 1u) / DW_T;        /* 16-byte pieces per chunk */
  const u32 cs = tid * ppc * 16u < n ? tid * ppc * 16u : n;
  const u32 ce = cs + ppc * 16u < n ? cs + ppc * 16u : n;
  {
    u32 f0 = 0, f1 = 1, f2 = 2, f3 = 3, f4 = 4;
    u32 pb = cs > 0u ? W[cs - 1u] : 256u;
    for (u32 i = cs; i < ce; i += 16u) {
      u32 wq[4];
      if (i + 16u <= ce) { const uint4 q = *reinterpret_cast<const uint4 *>(W + i); wq[0] = q.x; wq[1] = q.y; wq[2] = q.z; wq[3] = q.w; }
      else for (u32 j = 0; j < 16u; j++) { if ((j & 3u) == 0u) wq[j >> 2] = 0; if (i + j < ce) wq[j >> 2] |= (u32)W[i + j] << (8u * (j & 3u)); }
      const u32 m = ce - i < 16u ? ce - i : 16u;
      for (u32 j = 0; j < m; j++) {
        const u32 b = (wq[j >> 2] >> (8u * (j & 3u))) & 255u;
        const bool eq = b == pb;
        f0 = rle_step(f0, eq); f1 = rle_step(f1, eq); f2 = rle_step(f2, eq); f3 = rle_step(f3, eq); f4 = rle_step(f4, eq);
        pb = b;
      }
    }
    S.fn[tid] = f0 | f1 << 3 | f2 << 6 | f3 << 9 | f4 << 12;
  }
  __syncthreads();
  if (tid == 0u) {                                    /* start state of every chunk (kept in fn[]) */
    u32 c = 0;
    for (u32 t = 0; t < DW_T; t++) { const u32 f = S.fn[t]; S.fn[t] = c; c = (f >> (3u * c)) & 7u; }
  }
  __syncthreads();
  /* decoded length, CRC from 0 and the per-piece records (offsets relative to the chunk for now) */
  u32 crc = 0, outl = 0;
  {
    u32 c = S.fn[tid];
    u32 pb = cs > 0u ? W[cs - 1u] : 256u;
    for (u32 i = cs; i < ce; i += 16u) {
      pinfo[i >> 4] = outl << 3 | c;
      u32 wq[4];
      if (i + 16u <= ce) { const uint4 q = *reinterpret_cast<const uint4 *>(W + i); wq[0] = q.x; wq[1] = q.y; wq[2] = q.z; wq[3] = q.w; }
      else for (u32 j = 0; j < 16u; j++) { if ((j & 3u) == 0u) wq[j >> 2] = 0; if (i + j < ce) wq[j >> 2] |= (u32)W[i + j] << (8u * (j & 3u)); }
      const u32 m = ce - i < 16u ? ce - i : 16u;
      for (u32 j = 0; j < m; j++) {
        const u32 b = (wq[j >> 2] >> (8u * (j & 3u))) & 255u;
        if (c == 4u) {                                /* a count: b more copies of the run's byte */
          for (u32 r = 0; r < b; r++) crc = dec_crc_step(S.crctab, crc, pb);
          outl += b;
          c = 0;
        } else {
          c = (c != 0u && b == pb) ? c + 1u : 1u;
          crc = dec_crc_step(S.crctab, crc, b);
          outl++;
        }
        pb = b;
      }
    }
    S.olen[tid] = outl;
  }
  __syncthreads();
  if (tid == 0u) {
    u32 acc = 0;
    for (u32 t = 0; t < DW_T; t++) { const u32 v = S.olen[t]; S.olen[t] = acc; acc += v; }
    S.total = acc;
    S.xr[0] = S.xr[1] = S.xr[2] = S.xr[3] = 0;
  }
  __syncthreads();
  const u32 total = S.total, mybase = S.olen[tid];
  for (u32 i = cs; i < ce; i += 16u) pinfo[i >> 4] += mybase << 3;
  /* 5. the block CRC */
  u32 term = crc_shift(S.pow8, crc, total - mybase - outl);
  if (tid == 0u) term ^= crc_shift(S.pow8, 0xFFFFFFFFu, total);
  for (u32 d = 32u; d >= 1u; d >>= 1) term ^= (u32)__shfl_xor((int)term, d);
  if ((tid & 63u) == 0u) S.xr[tid >> 6] = term;
  __syncthreads();
  if (tid == 0u) {
    const u32 cc = ~(S.xr[0] ^ S.xr[1] ^ S.xr[2] ^ S.xr[3]);
    D->computed_crc = cc;
    D->out_len = total;
    if (cc != D->stored_crc) D->err = 11;
  }
}

/* ------------------------------------------------------------------ k_demit */
/* inverse RLE1 of W[] to out + out_off: every lane owns 16 bytes of W and starts from the record k_dwalk
 * left for them (output offset << 3 | state); neighbouring lanes write neighbouring bytes.            */
#define DE_SPLIT 8u
__global__ void __launch_bounds__(256)
k_demit(const lbz_dblock *blocks, u32 nblk, const u8 *W_base, const u32 *pinfo_base, u8 *out, u64 out_cap, u32 cap)
{
  const u32 tid = threadIdx.x;
  const u32 blk = blockIdx.x / DE_SPLIT, part = blockIdx.x % DE_SPLIT;
  if (blk >= nblk) return;
  const lbz_dblock *D = &blocks[blk];
  const u32 n = D->nblock;
  if (D->err || n == 0u) return;
  if (D->out_off + D->out_len > out_cap) return;
  const u8 *W = W_base + (size_t)blk * cap;
  const u32 *pinfo = pinfo_base + (size_t)blk * (cap / 16u);
  u8 *ob = out + D->out_off;
  const u32 npiece = (n + 15u) / 16u;
  const u32 per = (npiece + DE_SPLIT - 1u) / DE_SPLIT;
  const u32 p0 = part * per, p1 = p0 + per < npiece ? p0 + per : npiece;
  for (u32 p = p0 + tid; p < p1; p += 256u) {
    const u32 i = p * 16u;
    const u32 rec = pinfo[p];
    u32 c = rec & 7u;
    u8 *o = ob + (rec >> 3);
    u32 wq[4];
    if (i + 16u <= n) { const uint4 q = *reinterpret_cast<const uint4 *>(W + i); wq[0] = q.x; wq[1] = q.y; wq[2] = q.z; wq[3] = q.w; }
    else for (u32 j = 0; j < 16u; j++) { if ((j & 3u) == 0u) wq[j >> 2] = 0; if (i + j < n) wq[j >> 2] |= (u32)W[i + j] << (8u * (j & 3u)); }
    u32 pb = i > 0u ? W[i - 1u] : 256u;
    const u32 m = n - i < 16u ? n - i : 16u;
    for (u32 j = 0; j < m; j++) {
      const u32 b = (wq[j >> 2] >> (8u * (j & 3u))) & 255u;
      if (c == 4u) {
        for (u32 r = 0; r < b; r++) *o++ = (u8)pb;
        c = 0;
      } else {
        c = (c != 0u && b == pb) ? c + 1u : 1u;
        *o++ = (u8)b;
      }
      pb = b;
    }
  }
}
