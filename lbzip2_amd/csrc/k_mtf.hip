/*
 * k_mtf.hip -- stage 3: move-to-front ranks, zero-run (RUNA/RUNB) coding and the symbol
 * histogram of one block, one block per workgroup.
 *
 * Replaces make_map_e() and do_mtf() (reference src/encode.c:340-355, 360-425).  The
 * reference walks a 255-entry list per byte; here the MTF rank of a symbol is computed as
 *
 *     rank(i) = #{ symbols s' : last occurrence of s' before i  >  last occurrence of c_i }
 *
 * with never-seen symbols ordered by their index (virtual positions -1-s).  Each wave owns
 * one contiguous slice of the BWT string; its 64 lanes hold the 256 "last seen at" positions
 * (4 per lane), a rank is 4 ballots + popcounts, and only run heads (c_i != c_{i-1}) need
 * one -- every other position has rank 0.  The slices' start states come from a per-slice
 * "last occurrence" table (LDS atomicMax) chained over the 16 slices.
 *
 * Zero runs: a maximal run of k rank-0 positions becomes the bijective base-2 digits of k
 * (encode.c:381-386); non-zero rank r becomes symbol r+1; EOB closes the block.  Output
 * offsets are a workgroup add-scan over positions, after a max-scan that tells each
 * non-zero position where the previous one was.
 *
 * Traffic: reads N_rle bytes twice (+1 B/pos of rank scratch), writes 2 B per MTF symbol.
 */
#include "lbz_common.h"
#undef LBZ_WG
#define LBZ_WG LBZ_MTF_WG
#undef LBZ_NW
#define LBZ_NW (LBZ_WG / 64)
#include "lbz_kernels.h"

#define MTF_IPT 16u
#define MTF_TILE (LBZ_WG * MTF_IPT)
#define MTF_CHUNK 1024u                 /* bytes of a slice staged in LDS at a time */

struct mtf_lds {
  wg_scratch sc;
  int last[LBZ_NW][256];       /* per-slice last occurrence, then slice start state */
  u32 hist[LBZ_MAX_ALPHA + 2];
  int front[256];              /* last occurrence of every code in front of the workgroup's range (k_mtf_ranks; -1: none) */
  u8 cmap[256];
  u8 slot_of[256];
  u32 bc[4];
  u8 rk[LBZ_NW][256];          /* move-to-front rank of every symbol (slot) in front of the wave's current strip */
  __attribute__((aligned(16))) u8 stage_in[LBZ_NW][MTF_CHUNK];
  __attribute__((aligned(16))) u8 stage_out[LBZ_NW][MTF_CHUNK];
  u8 heads[LBZ_NW][MTF_CHUNK];   /* mtf_ranks: the codes of a chunk's run heads side by side, then their ranks */
  u64 hmask[LBZ_NW][MTF_CHUNK / 64u];   /* ... and which positions of the chunk are heads, a word a strip */
};

__device__ __forceinline__ u32 zrun_digits(u32 z)          /* floor(log2(z+1)) */
{
  return 31u - (u32)__clz(z + 1u);
}

/* MTF ranks of one wave's slice [lo, hi), a strip of 64 positions at a time, every lane its own position -- no loop over
 * the run heads (round 3; the loop it replaces cost 22 dependent scalar instructions per head on text and 40 on
 * high-entropy data, where every position is a head: 4.4 x the time of text on random bytes).
 *
 * State: rk[s] = move-to-front rank of symbol s in front of the strip (LDS, one byte-sized entry per symbol and wave).
 * For lane l with symbol c_l:
 *   prev_l   = the nearest lane below l with the same symbol (match-any ballots), none: first occurrence in the strip
 *   A_l      = distinct symbols in lanes (prev_l, l)  -- every one of them was used after c_l's last use.  A lane stays
 *              "alive" until its symbol comes again; the lanes killed in front of l are a prefix-OR over the lanes of
 *              one bit each (DPP scan), and A_l is a popcount of alive lanes above prev_l
 *   B_l      = (first occurrences only) symbols that do NOT occur in lanes [0, l) and were used after c_l's last use
 *            = rk[c_l] - #{first occurrences l' < l with rk[c_l'] < rk[c_l]}: the ranks taken by the strip's first
 *              occurrences in front of l are a prefix-OR of one-hot rank vectors (NQ 64-bit words), the count a masked popcount
 *   rank_l   = A_l + B_l, kept for run heads only (every other position has rank 0).
 * Behind the strip: symbols that occurred take the ranks 0.. in the order of their last occurrence (alive lanes, from
 * the right); every other symbol moves back by the number of strip symbols that stood behind it (popcount of the strip's
 * rank set above its rank).  NQ = 64-symbol words the alphabet needs (1, 2 or 4).
 * The slice is staged through LDS 1 KB at a time (one 16-byte load per lane, requested a chunk ahead; ranks leave the
 * same way).                                                                                                          */
template <int NQ>
__device__ __forceinline__ void mtf_ranks(const u8 *bwt, u8 *rk_out, u32 lo, u32 hi, mtf_lds *S)
{
  const u32 lane = lane_id(), w = wave_id();
  u8 *rk = S->rk[w];
  {
    /* ranks in front of the slice from the "last seen at" positions (k_mtf's prelude): rank = symbols seen later */
    const int *L = S->last[w];
    int mine[NQ];
    u32 cnt[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { mine[q] = L[lane + 64u * q]; cnt[q] = 0; }
    for (u32 t = 0; t < 64u * NQ; t++) {
      const int o = L[t];
#pragma unroll
      for (int q = 0; q < NQ; q++) cnt[q] += o > mine[q] ? 1u : 0u;
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < NQ; q++) rk[lane + 64u * q] = (u8)cnt[q];
    wave_sync();
  }
  u8 *inb = S->stage_in[w], *outb = S->stage_out[w];
  int carry = lo > 0u ? (int)S->cmap[bwt[lo - 1u]] : -1;     /* code of the position before the strip */
  const u64 below = lanes_below();

  uint4 nxt = { 0u, 0u, 0u, 0u };
  if (lo + 16u * lane + 16u <= hi) nxt = *reinterpret_cast<const uint4 *>(bwt + lo + 16u * lane);
  for (u32 c0 = lo; c0 < hi; c0 += MTF_CHUNK) {
    const u32 q0 = c0 + 16u * lane;
    if (q0 + 16u <= hi) *reinterpret_cast<uint4 *>(inb + 16u * lane) = nxt;
    else for (u32 i = 0; i < 16u; i++) inb[16u * lane + i] = (q0 + i < hi) ? bwt[q0 + i] : (u8)0;
    const u32 qn = q0 + MTF_CHUNK;
    if (qn + 16u <= hi) nxt = *reinterpret_cast<const uint4 *>(bwt + qn);
    wave_sync();
    const u32 left = hi - c0;
    const u32 nstrip = left >= MTF_CHUNK ? MTF_CHUNK / 64u : (left + 63u) / 64u;
    /* RUN HEADS ONLY (round 6).  A position that repeats the symbol in front of it has rank 0 and changes nothing in the list, so
       the chain of ranks runs over the run heads alone -- a third of the positions of a text block's BWT, a seventh of a source
       tree's, though four strips in five hold at least one (the strip-at-a-time form did its whole work for each of those).
       Three passes over the chunk: (1) codes, head flags (kept, a 64-bit word a strip) and the heads' codes side by side in `cb`;
       (2) the ranks of the heads, 64 a strip, written over their codes; (3) every position takes its rank -- a head the next
       of `cb`, anything else 0.  wiki: k_mtf 10.1 -> 7.4 ms per 10^9 bytes, Python sources 10.1 -> 7.0, real tar 7.0 -> 5.6; random
       bytes, all heads, pay the two light passes: 14.2 -> 15.3 (profiles/r06_zz_mtf_heads2.txt).  (A variant that took the strips
       of a chunk of nearly all heads as they stand cost eight more scalar registers: past the 80 that let two workgroups share a
       CU.  And at amdgpu_num_sgpr(96) -- "occupancy 8" to the compiler -- the kernel computed WRONG ranks on the device, differently
       from run to run: the 80 of the attribute below are a limit of the hardware, not a tuning.) */
    u8 *cb = S->heads[w];
    u64 *hmw = S->hmask[w];
    u32 nh = 0;
    for (u32 t = 0; t < nstrip; t++) {
      const u32 p = c0 + 64u * t + lane;
      const bool ok = p < hi;
      const int c = ok ? (int)S->cmap[inb[64u * t + lane]] : 0;
      int cprev = wave_shr1(c);
      if (lane == 0u) cprev = carry;
      carry = __builtin_amdgcn_readlane(c, 63);
      const bool head = ok && c != cprev;
      const u64 hm = __ballot(head);
      if (lane == 0u) hmw[t] = hm;
      if (head) cb[nh + (u32)__popcll(hm & below)] = (u8)c;
      nh += (u32)__popcll(hm);
    }
    wave_sync();
    const u32 nwork = (nh + 63u) / 64u;
    for (u32 t = 0; t < nwork; t++) {
      const bool ok = 64u * t + lane < nh, head = ok;
      const int c = ok ? (int)cb[64u * t + lane] : 0;
      const u64 okm = __ballot(ok);
      /* lanes with my symbol; the nearest one below me */
      const u64 mm = match_digit((u32)c, ok);                      /* (ballots: every lane takes part) */
      const u64 pm = ok ? mm & below : 0ull;
      const bool first = ok && pm == 0ull;
      const u32 prev = pm ? 63u - (u32)__clzll((long long)pm) : 0u;
      /* a lane is killed by the next lane with its symbol: the lanes killed in front of me */
      u64 kall;
      const u64 killed = wave_excl_or64(pm ? 1ull << prev : 0ull, &kall);
      const u64 alive = below & okm & ~killed;
      const u32 A = (u32)__popcll(pm ? alive & ~((2ull << prev) - 1ull) : alive);
      /* first occurrences: my symbol's rank in front of the strip, minus the first occurrences in front of me that stood before it */
      const u32 r0 = ok ? (u32)rk[c] : 0u;
      u32 B = r0;
      u64 mtot[NQ];
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const bool inq = first && (r0 >> 6) == (u32)q;
        if (NQ > 1 && q > 0 && __ballot(inq) == 0ull) { mtot[q] = 0ull; continue; }   /* no first occurrence from this far back: usual for q >= 1 */
        const u64 mine = inq ? 1ull << (r0 & 63u) : 0ull;
        const u64 seen = wave_excl_or64(mine, &mtot[q]);           /* ranks taken by first occurrences in front of me, word q */
        const u64 lowbits = (r0 >> 6) > (u32)q ? ~0ull : ((r0 >> 6) == (u32)q ? (1ull << (r0 & 63u)) - 1ull : 0ull);
        B -= (u32)__popcll(seen & lowbits);
      }
      const u32 myrank = head ? A + (first ? B : 0u) : 0u;
      if (ok) cb[64u * t + lane] = (u8)myrank;                     /* (every lane of the strip has read its code by now) */
      /* the list behind the strip */
      wave_sync();
      u32 above[NQ];                                               /* strip symbols with ranks in the words above word q */
      {
        u32 run = 0;
#pragma unroll
        for (int q = NQ - 1; q >= 0; q--) { above[q] = run; run += (u32)__popcll(mtot[q]); }
      }
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const u32 r = rk[lane + 64u * q];
        const u32 wq = r >> 6, bq = r & 63u;
        u64 mw = mtot[0];
        u32 ab = above[0];
#pragma unroll
        for (int j = 1; j < NQ; j++) if (wq == (u32)j) { mw = mtot[j]; ab = above[j]; }
        const u32 behind = ab + (u32)__popcll(bq == 63u ? 0ull : mw >> (bq + 1u));      /* strip symbols that stood behind this one */
        rk[lane + 64u * q] = (u8)(r + behind);                     /* (entries of the strip's own symbols are rewritten below) */
      }
      wave_sync();
      const u64 alive_end = okm & ~kall;                           /* last occurrence of every strip symbol */
      if (ok && ((alive_end >> lane) & 1ull)) rk[c] = (u8)__popcll(alive_end & ~((2ull << lane) - 1ull));
      wave_sync();
    }
    wave_sync();
    {                                                             /* (3) the ranks back to their positions */
      u32 base = 0;
      for (u32 t = 0; t < nstrip; t++) {
        const u64 hm = hmw[t];
        const u32 r = ((hm >> lane) & 1ull) ? (u32)cb[base + (u32)__popcll(hm & below)] : 0u;
        outb[64u * t + lane] = (u8)r;
        base += (u32)__popcll(hm);
      }
    }
    wave_sync();
    if (q0 + 16u <= hi) *reinterpret_cast<uint4 *>(rk_out + q0) = *reinterpret_cast<const uint4 *>(outb + 16u * lane);
    else for (u32 i = 0; i < 16u; i++) if (q0 + i < hi) rk_out[q0 + i] = outb[16u * lane + i];
    wave_sync();
  }
}

/* Stage 1 of a block: dense codes, the slices' start states, the ranks of the run heads -> rk[].  The block's positions are
 * dealt over `parts` workgroups (1: the whole block, k_mtf; 2 or 4: k_mtf_ranks, rounds of fewer blocks than the device has
 * CUs): workgroup `part` owns LBZ_NW consecutive slices, one per wave.  A slice starts from the last occurrences of every
 * symbol in front of it: those inside the workgroup's own range are chained over its slices as before; those in front of the
 * range come from one more scan of bwt[0, range start) by the whole workgroup -- the price of not talking to the other
 * workgroups (a quarter of the prelude's time per part: the scan touches LDS only at the ends of runs).  Ranks do not depend
 * on the slot numbering, so every workgroup numbers by its own head counts.  Returns the number of bytes in use. */
__device__ __forceinline__ u32 mtf_rank_stage(const u8 *bwt, u8 *rk, const lbz_block_meta *M, u32 n, u32 part, u32 parts, mtf_lds &S)
{
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  /* dense symbol numbering of the used bytes (encode.c:340-355) */
  u32 tot_inuse;
  {
    const u32 f = (tid < 256u && M->inuse[tid]) ? 1u : 0u;
    const u32 ex = wg_excl_add(f, &tot_inuse, &S.sc);
    if (tid < 256u) S.cmap[tid] = (u8)ex;
  }
  for (u32 i = tid; i < LBZ_NW * 256u; i += LBZ_WG) (&S.last[0][0])[i] = -1;
  for (u32 i = tid; i < LBZ_MAX_ALPHA + 2u; i += LBZ_WG) S.hist[i] = 0;
  if (tid < 256u) S.front[tid] = -1;
  __syncthreads();

  /* slices: one per wave, multiples of 64 positions */
  const u32 nsl = LBZ_NW * parts;
  const u32 cs = (((n + nsl - 1u) / nsl) + 63u) & ~63u;
  const u32 sl = part * LBZ_NW + w;
  const u32 lo = (u64)sl * cs < n ? sl * cs : n;
  const u32 hi = lo + cs < n ? lo + cs : n;
  const u32 wg_lo = (u64)part * LBZ_NW * cs < n ? part * LBZ_NW * cs : n;

  /* last occurrences in front of the workgroup's range (parts > 1): every thread four positions a trip, run ends only */
  for (u32 b0 = 256u * (tid >> 6); b0 < wg_lo; b0 += 256u * LBZ_NW) {
    u32 by[4];
#pragma unroll
    for (u32 k = 0; k < 4u; k++) { const u32 p = b0 + 64u * k + lane; by[k] = p < wg_lo ? bwt[p] : 0u; }
#pragma unroll
    for (u32 k = 0; k < 4u; k++) {
      const u32 p = b0 + 64u * k + lane;
      const u32 after = lane_from_above(by[k]);
      if (p < wg_lo && (lane == 63u || p + 1u == wg_lo || after != by[k])) atomicMax(&S.front[S.cmap[by[k]]], (int)p);
    }
  }
  for (u32 b0 = lo; b0 < hi; b0 += 256u) {
    u32 by[4];
#pragma unroll
    for (u32 k = 0; k < 4u; k++) { const u32 p = b0 + 64u * k + lane; by[k] = p < hi ? bwt[p] : 0u; }
#pragma unroll
    for (u32 k = 0; k < 4u; k++) {
      const u32 p = b0 + 64u * k + lane;
      const u32 before = lane_from_below(by[k]);
      const u32 after = lane_from_above(by[k]);
      if (p < hi) {
        /* only the ends of runs touch LDS: far fewer atomics, and fewer of them on one address */
        const u32 code = S.cmap[by[k]];
        if (lane == 63u || p + 1u == hi || after != by[k]) atomicMax(&S.last[w][code], (int)p);
        if (lane == 0u || before != by[k]) atomicAdd(&S.hist[code], 1u);    /* run heads (about) */
      }
    }
  }
  __syncthreads();
  /* Slots: the symbols numbered by falling head count, so that register 0 of mtf_ranks holds
     the 64 busiest.  Ranks do not depend on the numbering; the initial order does (never-seen
     symbols rank by code, virtual positions -1-code).                                       */
  u32 slot = tid;
  int st[LBZ_NW];
  if (tid < 256u) {
    if (tid < tot_inuse) {
      const u32 mine = S.hist[tid];
      u32 r = 0;
      for (u32 j = 0; j < tot_inuse; j++) { const u32 o = S.hist[j]; r += (o > mine || (o == mine && j < tid)) ? 1u : 0u; }
      slot = r;
    }
    int run = S.front[tid] >= 0 ? S.front[tid] : -1 - (int)tid;
#pragma unroll
    for (u32 w2 = 0; w2 < LBZ_NW; w2++) {
      const int t = S.last[w2][tid];
      st[w2] = run;
      if (t >= 0) run = t;
    }
  }
  __syncthreads();
  if (tid < 256u) {
#pragma unroll
    for (u32 w2 = 0; w2 < LBZ_NW; w2++) S.last[w2][slot] = st[w2];
    S.slot_of[tid] = (u8)slot;
  }
  __syncthreads();
  if (tid < 256u) S.cmap[tid] = S.slot_of[S.cmap[tid]];       /* byte -> slot */
  __syncthreads();

  /* ranks at run heads; NQ = 64-symbol words the alphabet needs */
  if (tot_inuse <= 64u) mtf_ranks<1>(bwt, rk, lo, hi, &S);
  else if (tot_inuse <= 128u) mtf_ranks<2>(bwt, rk, lo, hi, &S);
  else mtf_ranks<4>(bwt, rk, lo, hi, &S);
  return tot_inuse;
}

/* Stage 2 of a block: zero-run coding of the ranks (encode.c:381-386), the symbol histogram, the end-of-block symbol and
 * the padding of the last group.  One workgroup per block: output offsets are a scan over the whole block. */
__device__ __forceinline__ void mtf_zrle_stage(const u8 *rk, u16 *mtfv, u32 *freq_out, u32 blk, lbz_block_meta *M, u32 n, u32 tot_inuse, mtf_lds &S)
{
  const u32 tid = threadIdx.x, lane = lane_id();
  const u32 eob = tot_inuse + 1u;
  u32 hot0 = 0, hot1 = 0, hot2 = 0;   /* RUNA, RUNB and rank 1 are counted in registers: half of all symbols */
  u32 carry_nz = 0;        /* (position of the last non-zero rank) + 1 */
  u32 o_base = 0;
  for (u32 t0 = 0; t0 < n; t0 += MTF_TILE) {
    const u32 p0 = t0 + tid * MTF_IPT;
    u8 r[MTF_IPT];
    if (p0 + MTF_IPT <= n && ((uintptr_t)(rk + p0) & 15u) == 0u) {
      const uint4 qv = *reinterpret_cast<const uint4 *>(rk + p0);
      const u32 wq[4] = { qv.x, qv.y, qv.z, qv.w };
#pragma unroll
      for (u32 i = 0; i < MTF_IPT; i++) r[i] = (u8)(wq[i >> 2] >> (8u * (i & 3u)));
    } else {
#pragma unroll
      for (u32 i = 0; i < MTF_IPT; i++) r[i] = (p0 + i < n) ? rk[p0 + i] : (u8)0;
    }
    u32 lastnz = 0;
#pragma unroll
    for (u32 i = 0; i < MTF_IPT; i++) if (r[i]) lastnz = p0 + i + 1u;
    u32 enz, d0, tnz, d1;
    wg_excl_max_add(lastnz, 0u, &enz, &d0, &tnz, &d1, &S.sc);
    u32 prev1 = enz > carry_nz ? enz : carry_nz;
    const u32 prev1_start = prev1;
    u32 nout = 0;
#pragma unroll
    for (u32 i = 0; i < MTF_IPT; i++) {
      if (r[i]) {
        const u32 z = p0 + i - prev1;
        nout += (z ? zrun_digits(z) : 0u) + 1u;
        prev1 = p0 + i + 1u;
      }
    }
    u32 ttot;
    u32 o = o_base + wg_excl_add(nout, &ttot, &S.sc);
    prev1 = prev1_start;
#pragma unroll
    for (u32 i = 0; i < MTF_IPT; i++) {
      if (r[i]) {
        u32 z = p0 + i - prev1;
        while (z) {
          const u32 d = (z - 1u) & 1u;
          mtfv[o++] = (u16)d;
          if (d) hot1++; else hot0++;
          z = (z - 1u) >> 1;
        }
        mtfv[o++] = (u16)(r[i] + 1u);
        if (r[i] == 1u) hot2++; else atomicAdd(&S.hist[r[i] + 1u], 1u);
        prev1 = p0 + i + 1u;
      }
    }
    o_base += ttot;
    carry_nz = tnz > carry_nz ? tnz : carry_nz;
  }
  hot0 = wave_sum(hot0); hot1 = wave_sum(hot1); hot2 = wave_sum(hot2);
  if (lane == 0u) { atomicAdd(&S.hist[0], hot0); atomicAdd(&S.hist[1], hot1); atomicAdd(&S.hist[2], hot2); }
  __syncthreads();
  if (tid == 0) {
    u32 o = o_base;
    u32 z = n - carry_nz;                        /* trailing zero run */
    while (z) {
      const u32 d = (z - 1u) & 1u;
      mtfv[o++] = (u16)d;
      S.hist[d]++;
      z = (z - 1u) >> 1;
    }
    mtfv[o++] = (u16)eob;
    S.hist[eob]++;
    const u32 nm = o;
    const u32 padded = (nm + LBZ_GROUP - 1u) / LBZ_GROUP * LBZ_GROUP;
    for (; o < padded; o++) mtfv[o] = (u16)(eob + 1u);      /* dummy symbol, encode.c:1034-1035 */
    M->nmtf = nm;
    M->alpha = eob + 1u;
  }
  __syncthreads();
  for (u32 i = tid; i < LBZ_MAX_ALPHA + 2u; i += LBZ_WG) freq_out[(size_t)blk * 260u + i] = S.hist[i];
}

/* two workgroups per CU: the SGPR file admits 8 waves per SIMD only at <= 80 SGPRs per wave */
__global__ void __launch_bounds__(LBZ_WG, 2) __attribute__((amdgpu_num_sgpr(80)))   /* (two a CU: 64 vector registers) */
k_mtf(const u8 *Bbase, u8 *Rbase, u16 *Vbase, u32 *freq_out, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs)
{
  __shared__ mtf_lds S;
  const u32 blk = lbz_round_block(first, count, blockIdx.x, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (n == 0u) return;
  const size_t off = lbz_elem_off(L, blk);
  const u8 *bwt = Bbase + off;
  u8 *rk = Rbase + off;
  u16 *mtfv = Vbase + off;               /* room for n + 1 + 50 symbols (cap >= M + 64) */
#ifdef MTF_TICKS
  const u64 tk0 = wall_clock64();
#endif
  const u32 tot_inuse = mtf_rank_stage(bwt, rk, M, n, 0u, 1u, S);
  __syncthreads();
#ifdef MTF_TICKS
  const u64 tk2 = wall_clock64();
#endif
  for (u32 i = threadIdx.x; i < LBZ_MAX_ALPHA + 2u; i += LBZ_WG) S.hist[i] = 0;
  __syncthreads();
  mtf_zrle_stage(rk, mtfv, freq_out, blk, M, n, tot_inuse, S);
#ifdef MTF_TICKS
  if (threadIdx.x == 0) { M->ticks[3] = 0; M->ticks[4] = (u32)(tk2 - tk0); M->ticks[5] = (u32)(wall_clock64() - tk2); }
#endif
}

/* The same in two launches, for rounds of fewer blocks than the device has CUs (round 5): the ranks with `parts` workgroups
 * per block (blockIdx = block * parts + part), then the zero-run coding with one.  A round of 112 blocks left half of the
 * CUs idle for the 2.3 ms a block's ranks take; as four workgroups they take a quarter of that (+ the scan of the text in
 * front of a part) and the second launch 0.5 ms. */
__global__ void __launch_bounds__(LBZ_WG) __attribute__((amdgpu_num_sgpr(80)))
k_mtf_ranks(const u8 *Bbase, u8 *Rbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs, u32 parts)
{
  __shared__ mtf_lds S;
  const u32 blk = lbz_round_block(first, count, blockIdx.x / parts, slabs);
  const lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (n == 0u) return;
  const size_t off = lbz_elem_off(L, blk);
  (void)mtf_rank_stage(Bbase + off, Rbase + off, M, n, blockIdx.x % parts, parts, S);
}

__global__ void __launch_bounds__(LBZ_WG) __attribute__((amdgpu_num_sgpr(80)))
k_mtf_zrle(const u8 *Rbase, u16 *Vbase, u32 *freq_out, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs)
{
  __shared__ mtf_lds S;
  const u32 tid = threadIdx.x;
  const u32 blk = lbz_round_block(first, count, blockIdx.x, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (n == 0u) return;
  const size_t off = lbz_elem_off(L, blk);
  u32 tot_inuse;
  {
    const u32 f = (tid < 256u && M->inuse[tid]) ? 1u : 0u;
    (void)wg_excl_add(f, &tot_inuse, &S.sc);
  }
  for (u32 i = tid; i < LBZ_MAX_ALPHA + 2u; i += LBZ_WG) S.hist[i] = 0;
  __syncthreads();
  mtf_zrle_stage(Rbase + off, Vbase + off, freq_out, blk, M, n, tot_inuse, S);
}
