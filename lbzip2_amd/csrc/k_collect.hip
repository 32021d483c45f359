/*
 * k_collect.hip -- stage 1 of the block compressor: initial run-length coding (RLE1),
 * block cutting, used-byte map and CRC-32, one workgroup per input slab.
 *
 * Replaces collect() (reference src/encode.c:135-336) plus the closing of an open run in
 * encode() (encode.c:443-447) and the slab re-queue loop of do_collect()
 * (compress.c:93-104).  The reference is a byte-serial state machine; here every input
 * position decides locally what it emits:
 *
 *   rs(p)  = start of the maximal run containing p (never before the block's first byte)
 *   kk     = (p - rs(p)) mod 259                  position inside its <=259-byte run chunk
 *   emits  = 1 byte  if kk < 3,   2 bytes (byte + count) if kk == 3,   nothing otherwise
 *   count  = min(255, equal bytes following p inside the slab)
 *
 * rs() is a workgroup max-scan of run-head positions, output offsets an add-scan of
 * `emits`.  The block is cut at the first position whose output would pass the block
 * capacity M (this reproduces encode.c:162-221 incl. the "never separate the 4th run byte
 * from its count" rule, :218); the rest of the slab becomes the slab's spill block, with
 * run positions restarted at the cut.  CRC-32 (poly 0x04C11DB7, MSB first, crctab.c) is
 * computed per thread over equal chunks and folded with x^(8*len) multiplications.
 *
 * Memory: reads the slab once (16 B per lane, coalesced), writes <= 1.25 bytes per input
 * byte; algorithmic traffic N_in + N_rle.
 */
#include "lbz_common.h"
#undef LBZ_WG
#define LBZ_WG LBZ_COLLECT_WG
#undef LBZ_NW
#define LBZ_NW (LBZ_WG / 64)
#include "lbz_kernels.h"

#define COL_IPT 16u
#define COL_TILE (LBZ_WG * COL_IPT)
#define CRC_POLY 0x04C11DB7u

struct collect_lds {
  wg_scratch sc;
  u32 crc_tab[256];
  u32 part[LBZ_WG];
  u32 inuse[256];
  u32 bc[4];
  /* one tile's output, laid out like the 16-byte vectors of the block array it goes to */
  __attribute__((aligned(16))) u8 stage[COL_TILE + COL_TILE / 4u + 64u];
};

__device__ __forceinline__ u32 crc_mulmod(u32 a, u32 b)
{
  u32 r = 0;
  for (int i = 31; i >= 0; i--) {
    r = (r << 1) ^ ((r >> 31) ? CRC_POLY : 0u);
    if ((b >> i) & 1u) r ^= a;
  }
  return r;
}

__device__ __forceinline__ u32 crc_xpow(u32 nbits)     /* x^nbits mod P */
{
  u32 r = 1u, b = 2u;
  while (nbits) {
    if (nbits & 1u) r = crc_mulmod(r, b);
    b = crc_mulmod(b, b);
    nbits >>= 1;
  }
  return r;
}

/* CRC (init 0xFFFFFFFF, no final inversion) of x[a..b).  All threads must call. */
__device__ __forceinline__ u32 wg_crc32(const u8 *x, u32 a, u32 b, u32 xlen, collect_lds *S)
{
  const u32 tid = threadIdx.x;
  const u32 len = b - a;
  const u32 C = (len + LBZ_WG - 1) / LBZ_WG;            /* bytes per thread */
  const u32 padn = C * LBZ_WG - len;                    /* virtual leading zero bytes */
  u32 crc = 0;
  {
    u32 v0 = tid * C, v1 = v0 + C;
    if (v1 > padn) {
      u32 p = a + (v0 > padn ? v0 - padn : 0u);
      const u32 pe = a + (v1 - padn);
      if (((uintptr_t)x & 15u) == 0u) {
        /* aligned 16-byte loads; bytes outside [p, pe) of the first/last vector are skipped */
        for (u32 base = p & ~15u; base < pe; base += 16u) {
          u32 wq[4];
          if (base + 16u <= xlen) {
            const uint4 v = *reinterpret_cast<const uint4 *>(x + base);
            wq[0] = v.x; wq[1] = v.y; wq[2] = v.z; wq[3] = v.w;
          } else {                                        /* never read past the input buffer */
#pragma unroll
            for (u32 k = 0; k < 4u; k++) {
              wq[k] = 0;
              for (u32 j = 0; j < 4u; j++) if (base + 4u * k + j < xlen) wq[k] |= (u32)x[base + 4u * k + j] << (8u * j);
            }
          }
#pragma unroll
          for (u32 i = 0; i < 16u; i++) {
            const u32 q = base + i;
            if (q >= p && q < pe) crc = (crc << 8) ^ S->crc_tab[(crc >> 24) ^ ((wq[i >> 2] >> (8u * (i & 3u))) & 255u)];
          }
        }
      } else {
        for (; p < pe; p++) crc = (crc << 8) ^ S->crc_tab[(crc >> 24) ^ x[p]];
      }
    }
  }
  S->part[tid] = crc;
  u32 xp = crc_xpow(8u * C);
  __syncthreads();
  for (u32 stride = 1; stride < LBZ_WG; stride <<= 1) {
    if ((tid & (2u * stride - 1u)) == 0u)
      S->part[tid] = crc_mulmod(S->part[tid], xp) ^ S->part[tid + stride];
    xp = crc_mulmod(xp, xp);
    __syncthreads();
  }
  u32 r = S->part[0] ^ crc_mulmod(0xFFFFFFFFu, crc_xpow(8u * len));
  __syncthreads();
  return r;
}

/* The count byte of a run (encode.c:247-275): how many bytes equal to `byte` follow from position q on, at most 255 and
 * not past `end`.  Byte by byte up to an 8-byte boundary, then eight at a time -- a byte-serial loop is 255 dependent
 * loads (127 us on a zero-filled slab, measured: the tile's other threads wait at the next barrier), this one 32.     */
__device__ __forceinline__ u32 run_count(const u8 *x, u32 q, u32 end, u8 byte, bool aligned_view)
{
  const u32 lim = end - q < 255u ? end - q : 255u;
  u32 c = 0;
  while (c < lim && ((q + c) & 7u) != 0u) { if (x[q + c] != byte) return c; c++; }
  if (aligned_view) {                                  /* x is 16-byte aligned: q + c is an 8-byte boundary of the allocation */
    const u64 pat = 0x0101010101010101ull * (u64)byte;
    while (c + 8u <= lim) {
      const u64 d = *reinterpret_cast<const u64 *>(x + q + c) ^ pat;
      if (d) return c + (u32)(__ffsll((long long)d) - 1) / 8u;
      c += 8u;
    }
  }
  while (c < lim && x[q + c] == byte) c++;
  return c;
}

/* Tokenise x[base..end) into out[] (capacity cap).  Returns via S->bc: [0] = bytes written,
 * [1] = first unconsumed position (== end if everything fitted).  EMIT = false: the same decisions without
 * the output (count bytes, staging, stores, used-byte map) -- where the block ends, nothing else.          */
template <bool EMIT = true>
__device__ __forceinline__ void collect_pass(const u8 *x, u32 base, u32 end, u32 cap, u8 *out, collect_lds *S,
                             u32 t_resume = 0xFFFFFFFFu, u32 o_resume = 0, u32 rs_resume = 0)
{
  const u32 tid = threadIdx.x;
  const bool vec_ok = ((uintptr_t)x & 15u) == 0u;
  u32 carry_rs = rs_resume;   /* (run start of the last byte seen) + 1; 0 = none yet */
  u32 o_base = o_resume;
  u32 cut = end;

  const u32 tfirst = t_resume != 0xFFFFFFFFu ? t_resume : base & ~(COL_TILE - 1u);   /* resume: the state seq_skip() left */
  uint4 nxt = { 0u, 0u, 0u, 0u };
  if (vec_ok && tfirst + tid * COL_IPT + COL_IPT <= end) nxt = *reinterpret_cast<const uint4 *>(x + tfirst + tid * COL_IPT);
  for (u32 t0 = tfirst; t0 < end; t0 += COL_TILE) {
    const u32 p0 = t0 + tid * COL_IPT;
    u8 b[COL_IPT];
    if (vec_ok && p0 + COL_IPT <= end) {
      const u32 wq[4] = { nxt.x, nxt.y, nxt.z, nxt.w };
#pragma unroll
      for (u32 i = 0; i < COL_IPT; i++) b[i] = (u8)(wq[i >> 2] >> (8u * (i & 3u)));
    } else {
#pragma unroll
      for (u32 i = 0; i < COL_IPT; i++) b[i] = (p0 + i < end) ? x[p0 + i] : (u8)0;
    }
    if (vec_ok && p0 + COL_TILE + COL_IPT <= end)        /* the next tile's bytes, requested a tile ahead */
      nxt = *reinterpret_cast<const uint4 *>(x + p0 + COL_TILE);
    /* the byte before this thread's first: the neighbour lane's last, or one load per wave */
    u8 prevb = (u8)lane_from_below((u32)b[COL_IPT - 1u]);
    if (lane_id() == 0u) prevb = (p0 > base && p0 <= end) ? x[p0 - 1] : (u8)0;

    /* run heads and the start of the run each position belongs to */
    u32 headmask = 0, lh = 0;
#pragma unroll
    for (u32 i = 0; i < COL_IPT; i++) {
      const u32 p = p0 + i;
      const bool act = p >= base && p < end;
      const bool head = act && (p == base || b[i] != (i ? b[i - 1] : prevb));
      if (head) { headmask |= 1u << i; lh = p + 1u; }
    }
    u32 emax, eadd_unused, tmax, tadd_unused;
    wg_excl_max_add(lh, 0u, &emax, &eadd_unused, &tmax, &tadd_unused, &S->sc);
    u32 rs = emax > carry_rs ? emax : carry_rs;

    u32 emit2 = 0;             /* 2 bits per position: bytes emitted */
    u8 cnt[COL_IPT];
    u32 nout = 0;
#pragma unroll
    for (u32 i = 0; i < COL_IPT; i++) {
      const u32 p = p0 + i;
      const bool act = p >= base && p < end;
      cnt[i] = 0;
      if (headmask & (1u << i)) rs = p + 1u;
      if (act) {
        const u32 kk = (p - (rs - 1u)) % LBZ_RUN_CAP;
        u32 e = kk < 3u ? 1u : (kk == 3u ? 2u : 0u);
        if (EMIT && kk == 3u) cnt[i] = (u8)run_count(x, p + 1u, end, b[i], vec_ok);
        emit2 |= e << (2u * i);
        nout += e;
      }
    }
    u32 ttot;
    u32 o = o_base + wg_excl_add(nout, &ttot, &S->sc);

    u32 limit = end;           /* positions >= limit are not consumed */
    if (o_base + ttot > cap) { /* the block fills inside this tile: find the cut */
      u32 cand = 0xFFFFFFFFu, oo = o, at = 0;
#pragma unroll
      for (u32 i = 0; i < COL_IPT; i++) {
        const u32 e = (emit2 >> (2u * i)) & 3u;
        if (cand == 0xFFFFFFFFu && e && oo + e > cap) { cand = p0 + i; at = oo; }
        oo += e;
      }
      limit = wg_min(cand, &S->sc);
      if (cand == limit) S->bc[0] = at;
      __syncthreads();
      cut = limit;
    }

    /* the tile's bytes go to LDS first and leave as whole 16-byte vectors of the block array */
    const u32 gbase = o_base & ~15u;
    u32 oo = o;
    if (EMIT) {
#pragma unroll
    for (u32 i = 0; i < COL_IPT; i++) {
      const u32 e = (emit2 >> (2u * i)) & 3u;
      if (e && p0 + i < limit) {
        S->stage[oo - gbase] = b[i];
        S->inuse[b[i]] = 1u;
        if (e == 2u) { S->stage[oo + 1u - gbase] = cnt[i]; S->inuse[cnt[i]] = 1u; }
      }
      oo += e;
    }
    }
    __syncthreads();
    if (EMIT) {
      const u32 o_end = (cut != end) ? S->bc[0] : o_base + ttot;        /* bytes of this block so far */
      const bool out_ok = ((uintptr_t)out & 15u) == 0u;
      for (u32 v = gbase + 16u * tid; v < o_end; v += 16u * LBZ_WG) {
        if (out_ok && v >= o_base && v + 16u <= o_end) {
          *reinterpret_cast<uint4 *>(out + v) = *reinterpret_cast<const uint4 *>(S->stage + (v - gbase));
        } else {
          for (u32 k = 0; k < 16u; k++) if (v + k >= o_base && v + k < o_end) out[v + k] = S->stage[v + k - gbase];
        }
      }
    }
    __syncthreads();
    if (cut != end) break;
    o_base += ttot;
    carry_rs = tmax > carry_rs ? tmax : carry_rs;
  }
  __syncthreads();
  if (tid == 0) {
    if (cut == end) S->bc[0] = o_base;
    S->bc[1] = cut;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(LBZ_WG, 4)
k_collect(const u8 *in, u64 in_len, lbz_layout L, u8 *Tbase, lbz_block_meta *meta, u32 first,
          const u32 *slabs, const u32 *slab_len)
{
  __shared__ collect_lds S;
  const u32 tid = threadIdx.x;
  const u32 slab = slabs ? slabs[blockIdx.x] : first + blockIdx.x;
  const u8 *x = in + (u64)slab * L.M;
  const u64 left = in_len - (u64)slab * L.M;
  const u32 len = slab_len ? slab_len[blockIdx.x] : (left < L.M ? (u32)left : L.M);   /* listed slabs carry their own length */

  if (tid < 256) {
    u32 c = tid << 24;
    for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ CRC_POLY : (c << 1);
    S.crc_tab[tid] = c;
  }

  u32 base = 0;
  for (u32 part = 0; part < 2; part++) {
    const u32 blk = 2u * slab + part;
    lbz_block_meta *m = &meta[blk];
    if (base >= len) {                       /* no spill block */
      if (tid == 0) { m->n = 0; m->consumed = 0; m->out_len = 0; m->err = 0; m->nmtf = 0; }
      continue;
    }
    if (tid < 256) S.inuse[tid] = 0;
    __syncthreads();
#ifdef COL_TICKS
    const u64 tk0 = wall_clock64();
#endif
    collect_pass(x, base, len, part ? L.cap_b : L.M, Tbase + lbz_elem_off(L, blk), &S);
    const u32 nblock = S.bc[0], stop = S.bc[1];
    __syncthreads();
#ifdef COL_TICKS
    const u64 tk1 = wall_clock64();
#endif
    const u32 crc = wg_crc32(x, base, stop, len, &S);
#ifdef COL_TICKS
    if (tid == 0) { m->ticks[6] = (u32)(tk1 - tk0); m->ticks[7] = (u32)(wall_clock64() - tk1); }
#endif
    if (tid < 256) m->inuse[tid] = (u8)S.inuse[tid];
    if (tid == 0) {
      m->n = nblock; m->crc = crc; m->consumed = stop - base;
      m->err = (part && stop != len) ? 1u : 0u;
      m->out_len = 0; m->nmtf = 0; m->periodic = 0; m->bwt_idx = 0;
    }
    base = stop;
    __syncthreads();
  }
}


/* The part of a block that cannot contain its end, in steps of 64 bytes per thread with one scan and one sum each: the same run
 * starts and emitted-byte counts as collect_pass, totals only.  Leaves the state collect_pass resumes from (tile,
 * bytes so far, run start carried) at the first step the block might fill in.                              */
#define SKIP_IPT 64u
#define SKIP_TILE (LBZ_WG * SKIP_IPT)
static_assert((SKIP_TILE & (SKIP_TILE - 1u)) == 0u, "steps are a power of two (tile tables of the sequential mode)");

/* a thread's 64 bytes at x + p0 (bytes at or behind `hi` read as zero), the byte before them (valid if p0 > xlo: xlo =
   first position of the view that may be read) and, bit i of eq: byte i equals the byte before it */
struct skip_regs { u32 w[SKIP_IPT / 4u]; u32 prevb; u64 eq; };
__device__ __forceinline__ void skip_load(const u8 *x, bool vec_ok, u32 p0, u32 xlo, u32 hi, skip_regs *r)
{
  if (vec_ok && p0 + SKIP_IPT <= hi) {
#pragma unroll
    for (u32 q = 0; q < SKIP_IPT / 16u; q++) {
      const uint4 v = *reinterpret_cast<const uint4 *>(x + p0 + 16u * q);
      r->w[4u * q] = v.x; r->w[4u * q + 1u] = v.y; r->w[4u * q + 2u] = v.z; r->w[4u * q + 3u] = v.w;
    }
  } else {
#pragma unroll
    for (u32 q = 0; q < SKIP_IPT / 4u; q++) {
      r->w[q] = 0;
      for (u32 j = 0; j < 4u; j++) if (p0 + 4u * q + j < hi) r->w[q] |= (u32)x[p0 + 4u * q + j] << (8u * j);
    }
  }
  u32 prevb = lane_from_below(r->w[SKIP_IPT / 4u - 1u] >> 24);
  if (lane_id() == 0u) prevb = (p0 > xlo && p0 <= hi) ? x[p0 - 1] : 0u;
  r->prevb = prevb;
  /* four bytes per step: zero bytes of w ^ (w shifted in by one) */
  u64 eq = 0;
  u32 pbyte = prevb;
#pragma unroll
  for (u32 q = 0; q < SKIP_IPT / 4u; q++) {
    const u32 xw = r->w[q] ^ ((r->w[q] << 8) | pbyte);
    const u32 nz = (((xw & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xw) & 0x80808080u;     /* 0x80 in every non-zero byte */
    const u32 y = (~nz & 0x80808080u) >> 7;
    eq |= (u64)((y | (y >> 7) | (y >> 14) | (y >> 21)) & 15u) << (4u * q);
    pbyte = r->w[q] >> 24;
  }
  r->eq = eq;
}

/* Bytes the thread's active positions emit: one each, one more where a run reaches its fourth byte (the count byte),
   none for the run's bytes after that -- counted on the bit masks.  A run that comes in from the left has `lead` bytes
   before me (rs = its start + 1) and t of mine; inside my range three continuing positions in a row mark the fourth
   byte of a run that began here.  Only a run that could pass its 259-byte chunk inside my range is walked byte by byte. */
__device__ __forceinline__ u32 skip_count(const skip_regs &r, u64 act, u64 heads, u32 p0, u32 rs)
{
  const u64 cont = act & ~heads;                      /* positions that continue the run of the byte before */
  const u32 first = act ? (u32)__ffsll((long long)act) - 1u : 0u;
  const bool leftrun = act && ((cont >> first) & 1ull);
  const u64 cf = cont >> first;
  const u32 t = leftrun ? (~cf ? (u32)__ffsll((long long)~cf) - 1u : 64u) : 0u;
  const u32 lead = leftrun ? p0 + first - (rs - 1u) : 0u;
  if (!leftrun || lead + t < LBZ_RUN_CAP) {
    const u64 tm = t >= 64u ? ~0ull : ((1ull << t) - 1ull) << first;
    const u64 inner = cont & ~tm;
    const u64 c3 = inner & (inner << 1) & (inner << 2);
    const u32 n3 = (u32)__popcll(c3 & ~(c3 << 1)), n4 = (u32)__popcll(c3) - n3;
    const u32 n3l = (leftrun && lead <= 3u && lead + t > 3u) ? 1u : 0u;
    const u32 n4l = (leftrun && lead + t > 4u) ? lead + t - (lead > 4u ? lead : 4u) : 0u;
    return (u32)__popcll(act) + n3 + n3l - n4 - n4l;
  }
  u32 k = 0, nout = 0;
  bool have = false;
#pragma unroll
  for (u32 i = 0; i < SKIP_IPT; i++) {
    if ((act >> i) & 1ull) {
      if ((heads >> i) & 1ull) { k = 0; have = true; }
      else if (!have) { k = (p0 + i - (rs - 1u)) % LBZ_RUN_CAP; have = true; }
      else { k++; if (k == LBZ_RUN_CAP) k = 0; }
      nout += k < 3u ? 1u : (k == 3u ? 2u : 0u);
    }
  }
  return nout;
}

__device__ __forceinline__ void seq_skip(const u8 *x, u32 base, u32 end, u32 cap, collect_lds *S, u32 *t_out, u32 *o_out, u32 *rs_out)
{
  const u32 tid = threadIdx.x;
  const bool vec_ok = ((uintptr_t)x & 15u) == 0u;
  u32 carry_rs = 0, o_base = 0;
  u32 t0 = base & ~(COL_TILE - 1u);
  for (; t0 < end; t0 += SKIP_TILE) {
    const u32 p0 = t0 + tid * SKIP_IPT;
    skip_regs r;
    skip_load(x, vec_ok, p0, base, end, &r);
    /* my positions inside [base, end), run heads among them (the block's first byte always is one) */
    const u32 nlo = base > p0 ? (base - p0 < SKIP_IPT ? base - p0 : SKIP_IPT) : 0u;
    const u32 nhi = end > p0 ? (end - p0 < SKIP_IPT ? end - p0 : SKIP_IPT) : 0u;
    const u64 mlo = nlo >= 64u ? ~0ull : (1ull << nlo) - 1ull, mhi = nhi >= 64u ? ~0ull : (1ull << nhi) - 1ull;
    const u64 act = mhi & ~mlo;
    u64 heads = act & ~r.eq;
    if (base >= p0 && base < p0 + SKIP_IPT) heads |= 1ull << (base - p0);
    heads &= act;
    const u32 lh = heads ? p0 + 64u - (u32)__clzll((long long)heads) : 0u;      /* (last head) + 1 */
    u32 emax, e_unused, tmax, t_unused;
    wg_excl_max_add(lh, 0u, &emax, &e_unused, &tmax, &t_unused, &S->sc);
    const u32 rs = emax > carry_rs ? emax : carry_rs;
    const u32 nout = skip_count(r, act, heads, p0, rs);
    const u32 total = wg_sum(nout, &S->sc);
    if (o_base + total > cap) break;                 /* the block may end in this step: collect_pass takes over here */
    o_base += total;
    carry_rs = tmax > carry_rs ? tmax : carry_rs;
  }
  *t_out = t0; *o_out = o_base; *rs_out = carry_rs;
}

/* ---- tables of the sequential mode: what a 32 KB step of the INPUT emits, whoever's block it falls into -------------
 * A block's tokens differ from the tokens of the input read as one piece only in the run its first byte sits in (a
 * block starts a run afresh).  So the step totals, their prefix sums and the run start carried into every step can
 * be computed for the whole input in parallel, ahead of the chain: a link of the chain then tokenises the stretch up
 * to the next step boundary, finds the step its block fills in by a search in the prefix sums and makes the exact cut
 * there -- two short passes instead of ~28 steps (k_collect_seq).  Steps are aligned to in16 = `in` rounded down to 16
 * bytes; positions below are indices from in16 (the input's first byte has index a0 = in - in16).
 *   k_seq_tiles   one workgroup per step: last and first run head inside, bytes emitted if the step began a run
 *   k_seq_prefix  one workgroup: the run start carried into each step, the totals corrected for the run that
 *                 continues into the step, exclusive prefix sums                                                  */
__device__ __forceinline__ u64 run_emits(u64 m)          /* bytes the first m bytes of a run emit: 4 + count per 259 */
{
  const u64 r = m % LBZ_RUN_CAP;
  return 5ull * (m / LBZ_RUN_CAP) + (r <= 3ull ? r : 5ull);
}

__global__ void __launch_bounds__(LBZ_WG, 4)
k_seq_tiles(const u8 *in, u64 in_len, u32 *last_head, u32 *first_head, u32 *emits)
{
  __shared__ collect_lds S;
  const u32 tid = threadIdx.x;
  const u32 a0 = (u32)((uintptr_t)in & 15u);
  const u64 tile = blockIdx.x;
  const u64 lo64 = tile * SKIP_TILE;                      /* index of the step's first position */
  const u64 total = (u64)a0 + in_len;
  if (lo64 >= total) return;
  /* view: the step's positions are [OFF, OFF + SKIP_TILE) of x, so that the byte before the step can be read */
  const u32 OFF = tile ? 16u : 0u;
  const u8 *x = in - a0 + lo64 - OFF;
  const u32 lo = OFF + (tile ? 0u : a0);                  /* the first step begins at the input's first byte */
  const u32 hi = OFF + (u32)(total - lo64 < SKIP_TILE ? total - lo64 : SKIP_TILE);
  const u32 p0 = OFF + tid * SKIP_IPT;
  skip_regs r;
  skip_load(x, true, p0, tile ? 0u : lo, hi, &r);
  const u32 nlo = lo > p0 ? (lo - p0 < SKIP_IPT ? lo - p0 : SKIP_IPT) : 0u;
  const u32 nhi = hi > p0 ? (hi - p0 < SKIP_IPT ? hi - p0 : SKIP_IPT) : 0u;
  const u64 mlo = nlo >= 64u ? ~0ull : (1ull << nlo) - 1ull, mhi = nhi >= 64u ? ~0ull : (1ull << nhi) - 1ull;
  const u64 act = mhi & ~mlo;
  u64 heads = act & ~r.eq;
  if (tile == 0 && lo >= p0 && lo < p0 + SKIP_IPT) heads |= 1ull << (lo - p0);     /* the input's first byte */
  heads &= act;
  const u32 lh = heads ? p0 + 64u - (u32)__clzll((long long)heads) : 0u;      /* (last head) + 1 */
  const u32 fh = heads ? p0 + (u32)__ffsll((long long)heads) - 1u : 0xFFFFFFFFu;
  u32 emax, e_unused, tmax, t_unused;
  wg_excl_max_add(lh, 0u, &emax, &e_unused, &tmax, &t_unused, &S.sc);
  const u32 rs = emax > lo + 1u ? emax : lo + 1u;         /* as if a run began with the step */
  const u32 nout = skip_count(r, act, heads, p0, rs);
  const u32 tot = wg_sum(nout, &S.sc);
  const u32 fmin = wg_min(fh, &S.sc);
  if (tid == 0) {
    last_head[tile] = tmax ? tmax - OFF : 0u;             /* (offset of the last head in the step) + 1, 0 = none */
    first_head[tile] = fmin == 0xFFFFFFFFu ? hi - lo : fmin - lo;   /* positions in front of the first head */
    emits[tile] = tot;
  }
}

__global__ void __launch_bounds__(LBZ_WG, 4)
k_seq_prefix(const u32 *last_head, const u32 *first_head, const u32 *emits, u32 ntiles, u32 a0,
             unsigned long long *carry, unsigned long long *gpre)
{
  __shared__ wg_scratch sc;
  const u32 tid = threadIdx.x;
  u32 carry_tile = 0;          /* (last step before this chunk that holds a head) + 1 */
  u64 run = 0;
  for (u32 k0 = 0; k0 < ntiles; k0 += LBZ_WG) {
    const u32 k = k0 + tid;
    const u32 lhd = k < ntiles ? last_head[k] : 0u;
    u32 emax, e_unused, tmax, t_unused;
    wg_excl_max_add(lhd ? k + 1u : 0u, 0u, &emax, &e_unused, &tmax, &t_unused, &sc);
    const u32 jj = emax > carry_tile ? emax : carry_tile;           /* step jj - 1 holds the head of the run that reaches step k */
    u32 g = 0;
    if (k < ntiles) {
      /* index of that head; the input's first byte (index a0, step 0) is one, so steps k >= 1 always have one */
      const u64 cpos = jj ? (u64)(jj - 1u) * SKIP_TILE + (last_head[jj - 1u] - 1u) : (u64)a0;
      carry[k] = cpos;
      const u64 t = first_head[k];                                   /* positions of step k that continue that run */
      g = emits[k];
      if (k > 0 && t > 0) {
        const u64 lead = (u64)k * SKIP_TILE - cpos;
        g = (u32)((u64)g - run_emits(t) + run_emits(lead + t) - run_emits(lead));
      }
    }
    u32 tot;
    const u32 ex = wg_excl_add(g, &tot, &sc);
    if (k < ntiles) gpre[k] = run + ex;
    run += tot;
    carry_tile = tmax > carry_tile ? tmax : carry_tile;
  }
  if (tid == 0) gpre[ntiles] = run;
}

/* ---- -u / --sequential (compress.c:129-198, do_collect_seq): a block takes input until it is full, whatever
 * the slab boundaries, so where block b starts is known only when block b - 1 has been cut.  One launch, one
 * workgroup per block slot; workgroups take their block number from a ticket counter (so every predecessor is
 * resident or done) and wait for the predecessor to publish its cut: the tokenising pass of the blocks runs as
 * a chain, the CRC of a block -- the longer part -- runs beside its successors' passes.
 * starts[b] = (input position of block b) + 1, 0 = not known yet; starts[0] comes from the host.            */
__global__ void __launch_bounds__(LBZ_WG, 4)
k_collect_seq(const u8 *in, u64 in_len, lbz_layout L, u8 *Tbase, lbz_block_meta *meta, u32 nblk,
              unsigned long long *starts, u32 *ticket, lbz_seq_out *so, u32 slot0,
              const unsigned long long *carry, const unsigned long long *gpre, u32 ntiles)   /* tables of k_seq_prefix, or null */
{
  __shared__ collect_lds S;
  __shared__ unsigned long long s_start;
  const u32 tid = threadIdx.x;
  if (tid == 0) S.bc[3] = atomicAdd(ticket, 1u);
  if (tid < 256) {
    u32 c = tid << 24;
    for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ CRC_POLY : (c << 1);
    S.crc_tab[tid] = c;
    S.inuse[tid] = 0;
  }
  __syncthreads();
  const u32 b = S.bc[3];
  if (b >= nblk) return;
  if (tid == 0) {
    unsigned long long v = 0;
    for (u32 spins = 0; spins < (1u << 26); spins++) {          /* bounded: a lost predecessor must not hang the device */
      v = __atomic_load_n(&starts[b], __ATOMIC_RELAXED);
      if (v) break;
      __builtin_amdgcn_s_sleep(8);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    s_start = v;
  }
  __syncthreads();
  const u32 slot = slot0 + b;                                     /* block b of the chain lives in slab slot slot0 + b */
  lbz_block_meta *m = &meta[2u * slot], *m2 = &meta[2u * slot + 1u];
  if (tid == 0) { m2->n = 0; m2->consumed = 0; m2->out_len = 0; m2->err = 0; m2->nmtf = 0; }
  if (s_start == 0ull) {                                         /* the chain broke */
    if (tid == 0) { so->err = 1u; m->n = 0; m->consumed = 0; m->out_len = 0; m->err = 1u; m->nmtf = 0; __atomic_store_n(&starts[b + 1u], 0ull, __ATOMIC_RELEASE); }
    return;
  }
  const u64 p = s_start - 1ull;
  if (p >= in_len) {                                             /* input used up by the blocks before */
    if (tid == 0) {
      m->n = 0; m->consumed = 0; m->out_len = 0; m->err = 0; m->nmtf = 0;
      __atomic_store_n(&starts[b + 1u], p + 1ull, __ATOMIC_RELEASE);
      if (b + 1u == nblk) so->next = p;
    }
    return;
  }
  /* an aligned view of the input: x + base = in + p.  A block takes at most M / 5 runs of 259 bytes.  With the step
     tables the view begins at the step the block starts in (steps are aligned to `in` rounded down to 16 bytes). */
  const u32 a0 = (u32)((uintptr_t)in & 15u);
  const u64 j = p + a0;
  const u64 kp = gpre ? j / SKIP_TILE : 0ull;
  const u32 mis = gpre ? (u32)(j - kp * SKIP_TILE) : (u32)(((uintptr_t)in + p) & 15u);
  const u8 *x = in + p - mis;
  const u64 left = in_len - p;
  const u64 maxraw = (u64)L.M * 52u + 1024u;
  const u32 end = mis + (u32)(left < maxraw ? left : maxraw);
  u32 t_res, o_res, rs_res;
  __builtin_amdgcn_s_setprio(3);            /* the chain's link goes first on its SIMDs; its neighbours are off the chain */
  /* the tables apply from the first true run head behind the block's first byte: it must come before the next step */
  const bool fast = gpre && end > SKIP_TILE && kp + 1u < ntiles && carry[kp + 1u] >= j;
  if (fast) {
    seq_skip(x, mis, SKIP_TILE, L.M, &S, &t_res, &o_res, &rs_res);        /* the stretch up to the step boundary: o_res bytes */
    /* out(k) = o_res + gpre[k] - gpre[kp + 1] bytes are emitted in front of step k; the block fills in the last step
       whose out(k) still fits -- a search over the steps that hold positions in front of `end` */
    const u64 g0 = gpre[kp + 1u];
    u64 klo = kp + 1u, khi = kp + (u64)((end - 1u) / SKIP_TILE);
    if (khi > (u64)ntiles - 1u) khi = (u64)ntiles - 1u;
    while (klo < khi) {
      const u64 span = khi - klo + 1u, step = (span + LBZ_WG - 1u) / LBZ_WG;
      const u64 k = klo + (u64)tid * step;
      const u32 okv = (k <= khi && (u64)o_res + (gpre[k] - g0) <= (u64)L.M) ? tid + 1u : 0u;
      const u32 best = wg_max(okv, &S.sc);                          /* >= 1: out(klo) fits */
      const u64 kb = klo + (u64)(best - 1u) * step;
      klo = kb;
      khi = kb + step - 1u < khi ? kb + step - 1u : khi;
      if (step == 1u) break;
    }
    t_res = (u32)(klo - kp) * SKIP_TILE;
    o_res = (u32)((u64)o_res + (gpre[klo] - g0));
    rs_res = (u32)(carry[klo] - kp * SKIP_TILE) + 1u;
  } else {
    seq_skip(x, mis, end, L.M, &S, &t_res, &o_res, &rs_res);    /* 32 KB steps up to where the block might end ... */
  }
  collect_pass<false>(x, mis, end, L.M, nullptr, &S, t_res, o_res, rs_res);   /* ... and the cut itself: the successor can start */
  const u32 stop = S.bc[1];
  if (tid == 0) {
    const u64 nx = p + (u64)(stop - mis);
    __atomic_store_n(&starts[b + 1u], nx + 1ull, __ATOMIC_RELEASE);
    if (b + 1u == nblk || nx >= in_len) so->next = nx;
    atomicAdd(&so->nblocks, 1u);
    if (fast) atomicAdd(&so->nfast, 1u);
  }
  __builtin_amdgcn_s_setprio(0);
  __syncthreads();
  collect_pass<true>(x, mis, stop, L.M, Tbase + lbz_elem_off(L, 2u * slot), &S);      /* off the chain: bytes, used-byte map, CRC */
  const u32 nblock = S.bc[0];
  __syncthreads();
  const u32 crc = wg_crc32(x, mis, stop, end, &S);
  if (tid < 256) m->inuse[tid] = (u8)S.inuse[tid];
  if (tid == 0) {
    m->n = nblock; m->crc = crc; m->consumed = stop - mis;
    m->err = 0; m->out_len = 0; m->nmtf = 0; m->periodic = 0; m->bwt_idx = 0;
  }
}
