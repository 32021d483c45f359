/*
 * k_bwt.hip -- stage 2: Burrows-Wheeler transform of the CYCLIC rotations of one block,
 * one block per workgroup, persistent workgroups pulling blocks from a queue.
 *
 * Replaces divbwt() (reference src/divbwt.c:1706-1726; its sort_typeBstar/sssort/trsort/
 * construct_BWT machinery, divbwt.c:1488-1699, is a serial induced-sorting design with no
 * data-parallel analogue).  The BWT byte string is mathematically unique, so any correct
 * rotation sorter reproduces it; this one is prefix doubling built from two workgroup
 * primitives:
 *
 *   1. an LSD radix sort of (64-bit key, 32-bit value) pairs over the workgroup's private
 *      arrays in HBM: per-digit histograms and per-wave digit counters live in LDS, ranks
 *      inside a wave come from 8 ballots per item ("match-any"), tiles are scattered in
 *      order so every pass is stable;
 *   2. tiled max/add scans that turn equal-key runs into groups, ranks and the compacted
 *      list of still-tied rows.
 *
 * Round 0 sorts all n rotations by their first 8 bytes.  Round r (depth h = 8,16,...) re-keys
 * only rows that are still tied with (current group << 20 | rank of the rotation h further on)
 * and sorts that list (40 significant bits -> 5 passes, constant digits skipped).  The loop
 * ends when every row is unique or h >= n; in the latter case the block is exactly periodic
 * (T = u^k), equal rows stay tied and the origin pointer is the smallest equal row (the
 * reference's choice among the k equal rows is an artefact of its unstable quicksort,
 * SURVEY.md 8a-4 -- documented divergence, identical BWT bytes).
 *
 * HBM per slot: 44 B per element (lbz_common.h).  Algorithmic traffic of the stage as priced
 * in SURVEY.md 8(d): read T (1) + write SA (4) + read SA (4) + gather T (1) + write BWT (1)
 * = 11 B per block byte; the sorter's real traffic is reported next to it by bench.py.
 */
#include "lbz_kernels.h"

#define SORT_IPT 4u
#define SORT_TILE (LBZ_WG * SORT_IPT)
#define RANK_BITS 20u                   /* n <= 900000 < 2^20 */

struct bwt_lds {
  wg_scratch sc;
  u32 hist[8][256];
  u32 wcnt[LBZ_NW][256];
  u32 dbase[256];
  u32 bc[4];
};

struct bwt_slot {
  u64 *k0, *k1;
  u32 *v0, *v1, *sufx, *grp, *pos, *sa, *isa;
};

__device__ __forceinline__ bwt_slot slot_carve(u8 *ws, u32 cap)
{
  bwt_slot s;
  u8 *p = ws;
  s.k0 = (u64 *)p; p += (size_t)cap * 8u;
  s.k1 = (u64 *)p; p += (size_t)cap * 8u;
  s.v0 = (u32 *)p; p += (size_t)cap * 4u;
  s.v1 = (u32 *)p; p += (size_t)cap * 4u;
  s.sufx = (u32 *)p; p += (size_t)cap * 4u;
  s.grp = (u32 *)p; p += (size_t)cap * 4u;
  s.pos = (u32 *)p; p += (size_t)cap * 4u;
  s.sa = (u32 *)p; p += (size_t)cap * 4u;
  s.isa = (u32 *)p;
  return s;
}

/* Stable LSD radix sort of m (key,value) pairs on key bits [0, nbits).  Input in (k0,v0);
 * returns 0 if the sorted result is in (k0,v0), 1 if in (k1,v1).                       */
__device__ u32 wg_radix_sort(u64 *k0, u32 *v0, u64 *k1, u32 *v1, u32 m, u32 nbits, bwt_lds *S)
{
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const u32 npass = (nbits + 7u) / 8u;

  for (u32 i = tid; i < 8u * 256u; i += LBZ_WG) (&S->hist[0][0])[i] = 0;
  __syncthreads();
  for (u32 i = tid; i < m; i += LBZ_WG) {
    const u64 key = k0[i];
    for (u32 p = 0; p < npass; p++) atomicAdd(&S->hist[p][(u32)(key >> (8u * p)) & 255u], 1u);
  }
  __syncthreads();

  u32 cur = 0;
  for (u32 p = 0; p < npass; p++) {
    const u32 shift = 8u * p;
    /* digit offsets; a digit shared by every key makes the pass a no-op */
    const u32 c = tid < 256u ? S->hist[p][tid] : 0u;
    u32 tot;
    const u32 ex = wg_excl_add(c, &tot, &S->sc);
    if (tid < 256u) S->dbase[tid] = ex;
    if (tid == 0) S->bc[0] = 0;
    __syncthreads();
    if (tid < 256u && c == m) S->bc[0] = 1;
    __syncthreads();
    if (S->bc[0]) continue;

    const u64 *kin = cur ? k1 : k0;
    const u32 *vin = cur ? v1 : v0;
    u64 *kout = cur ? k0 : k1;
    u32 *vout = cur ? v0 : v1;

    for (u32 t0 = 0; t0 < m; t0 += SORT_TILE) {
      for (u32 i = tid; i < LBZ_NW * 256u; i += LBZ_WG) (&S->wcnt[0][0])[i] = 0;
      __syncthreads();
      u64 key[SORT_IPT];
      u32 val[SORT_IPT], rnk[SORT_IPT];
      const u32 wbase = t0 + w * 64u * SORT_IPT;
#pragma unroll
      for (u32 k = 0; k < SORT_IPT; k++) {
        const u32 i = wbase + k * 64u + lane;
        key[k] = i < m ? kin[i] : 0ull;
        val[k] = i < m ? vin[i] : 0u;
      }
#pragma unroll
      for (u32 k = 0; k < SORT_IPT; k++) {
        const u32 i = wbase + k * 64u + lane;
        const bool ok = i < m;
        const u32 d = (u32)(key[k] >> shift) & 255u;
        u64 mask = __ballot(ok);
#pragma unroll
        for (u32 b = 0; b < 8u; b++) {
          const bool bit = (d >> b) & 1u;
          const u64 bal = __ballot(bit);
          mask &= bit ? bal : ~bal;
        }
        const u32 below = (u32)__popcll(mask & lanes_below());
        const u32 prev = ok ? S->wcnt[w][d] : 0u;
        wave_sync();
        if (ok && below == 0u) S->wcnt[w][d] = prev + (u32)__popcll(mask);
        wave_sync();
        rnk[k] = prev + below;
      }
      __syncthreads();
      if (tid < 256u) {
        u32 run = S->dbase[tid];
#pragma unroll
        for (u32 w2 = 0; w2 < LBZ_NW; w2++) {
          const u32 t = S->wcnt[w2][tid];
          S->wcnt[w2][tid] = run;
          run += t;
        }
        S->dbase[tid] = run;
      }
      __syncthreads();
#pragma unroll
      for (u32 k = 0; k < SORT_IPT; k++) {
        const u32 i = wbase + k * 64u + lane;
        if (i < m) {
          const u32 d = (u32)(key[k] >> shift) & 255u;
          const u32 dst = S->wcnt[w][d] + rnk[k];
          kout[dst] = key[k];
          vout[dst] = val[k];
        }
      }
      __syncthreads();
    }
    cur ^= 1u;
  }
  return cur;
}

/* Turn a key-sorted list into groups.  For list entry k (row pos_k of the suffix array):
 *   head  = key differs from the previous entry
 *   rank  = row of the group's first entry
 * writes sa[row] = suffix, isa[suffix] = rank, and compacts the entries that are still
 * tied into (sufx, grp, pos).  pin == nullptr means row == k (round 0).  Returns the
 * number of still-tied entries.                                                         */
__device__ u32 wg_regroup(const u64 *key, const u32 *val, const u32 *pin, u32 m,
                          bwt_slot s, bwt_lds *S)
{
  const u32 tid = threadIdx.x;
  u32 carry_rank = 0, carry_cnt = 0;
  for (u32 t0 = 0; t0 < m; t0 += SORT_TILE) {
    const u32 k0 = t0 + tid * SORT_IPT;
    u64 kk[SORT_IPT + 2];
    u32 vv[SORT_IPT], row[SORT_IPT];
#pragma unroll
    for (u32 i = 0; i < SORT_IPT; i++) {
      const u32 k = k0 + i;
      kk[i + 1] = k < m ? key[k] : 0ull;
      vv[i] = k < m ? val[k] : 0u;
      row[i] = k < m ? (pin ? pin[k] : k) : 0u;
    }
    kk[0] = (k0 > 0 && k0 <= m) ? key[k0 - 1] : 0ull;
    kk[SORT_IPT + 1] = (k0 + SORT_IPT < m) ? key[k0 + SORT_IPT] : 0ull;

    u32 headmask = 0, lastrank = 0, nact = 0;
#pragma unroll
    for (u32 i = 0; i < SORT_IPT; i++) {
      const u32 k = k0 + i;
      if (k < m && (k == 0 || kk[i + 1] != kk[i])) { headmask |= 1u << i; lastrank = row[i] + 1u; }
    }
    /* entry k is still tied unless it and its successor both start a group */
    u32 actmask = 0;
#pragma unroll
    for (u32 i = 0; i < SORT_IPT; i++) {
      const u32 k = k0 + i;
      if (k < m) {
        const bool h = (headmask >> i) & 1u;
        const bool hn = (k + 1u >= m) || (kk[i + 2] != kk[i + 1]);
        if (!(h && hn)) { actmask |= 1u << i; nact++; }
      }
    }
    u32 erank, eact, trank, tact;
    wg_excl_max_add(lastrank, nact, &erank, &eact, &trank, &tact, &S->sc);
    u32 rank1 = erank > carry_rank ? erank : carry_rank;     /* (rank + 1) of the open group */
    u32 o = carry_cnt + eact;
#pragma unroll
    for (u32 i = 0; i < SORT_IPT; i++) {
      const u32 k = k0 + i;
      if (k < m) {
        if ((headmask >> i) & 1u) rank1 = row[i] + 1u;
        s.sa[row[i]] = vv[i];
        s.isa[vv[i]] = rank1 - 1u;
        if ((actmask >> i) & 1u) {
          s.sufx[o] = vv[i];
          s.grp[o] = rank1 - 1u;
          s.pos[o] = row[i];
          o++;
        }
      }
    }
    carry_rank = trank > carry_rank ? trank : carry_rank;
    carry_cnt += tact;
    /* the compacted columns are written at indices <= k of this tile, read only by later
       tiles: the barriers inside the scan already separate this tile's reads from writes */
  }
  __syncthreads();
  return carry_cnt;
}

__device__ void bwt_block(const u8 *T, u32 n, u8 *bwt, lbz_block_meta *meta, bwt_slot s, bwt_lds *S)
{
  const u32 tid = threadIdx.x;
  if (n == 1u) {                              /* divbwt.c:1712 */
    if (tid == 0) { bwt[0] = T[0]; meta->bwt_idx = 0; meta->periodic = 0; meta->rounds = 0; meta->sort_elems = 0; }
    __syncthreads();
    return;
  }

  /* round 0: first 8 bytes of every rotation, big-endian */
  for (u32 i = tid; i < n; i += LBZ_WG) {
    u64 key = 0;
    if (i + 8u <= n) {
#pragma unroll
      for (u32 k = 0; k < 8u; k++) key = (key << 8) | T[i + k];
    } else {
      u32 j = i;
      for (u32 k = 0; k < 8u; k++) { key = (key << 8) | T[j]; j = (j + 1u == n) ? 0u : j + 1u; }
    }
    s.k0[i] = key;
    s.v0[i] = i;
  }
  __syncthreads();
  u32 which = wg_radix_sort(s.k0, s.v0, s.k1, s.v1, n, 64u, S);
  u32 m = wg_regroup(which ? s.k1 : s.k0, which ? s.v1 : s.v0, nullptr, n, s, S);
  u32 rounds = 0, work = n;

  for (u32 h = 8u; m > 0u && h < n; h <<= 1) {
    /* re-key the tied rows: (group, rank of the rotation h bytes further on) */
    for (u32 k = tid; k < m; k += LBZ_WG) {
      const u32 sfx = s.sufx[k];
      u32 t = sfx + h;
      if (t >= n) t -= n;
      s.k0[k] = ((u64)s.grp[k] << RANK_BITS) | (u64)s.isa[t];
      s.v0[k] = sfx;
    }
    __syncthreads();
    which = wg_radix_sort(s.k0, s.v0, s.k1, s.v1, m, 2u * RANK_BITS, S);
    work += m;
    m = wg_regroup(which ? s.k1 : s.k0, which ? s.v1 : s.v0, s.pos, m, s, S);
    rounds++;
  }

  for (u32 j = tid; j < n; j += LBZ_WG) {
    const u32 sfx = s.sa[j];
    bwt[j] = T[sfx ? sfx - 1u : n - 1u];
  }
  if (tid == 0) {
    meta->bwt_idx = s.isa[0];
    meta->periodic = m > 0u ? 1u : 0u;
    meta->rounds = rounds;
    meta->sort_elems = work;
  }
  __syncthreads();
}

/* grid = number of workspace slots (persistent workgroups).  Queue order: all primary
 * blocks first (the big ones), then the spill blocks.                                    */
__global__ void __launch_bounds__(LBZ_WG)
k_bwt(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L,
      u32 nslabs, u32 *queue, u8 *ws, u64 slot_bytes)
{
  __shared__ bwt_lds S;
  const bwt_slot s = slot_carve(ws + (u64)blockIdx.x * slot_bytes, L.cap_a);
  for (;;) {
    if (threadIdx.x == 0) S.bc[1] = atomicAdd(queue, 1u);
    __syncthreads();
    const u32 q = S.bc[1];
    __syncthreads();
    if (q >= 2u * nslabs) break;
    const u32 blk = q < nslabs ? 2u * q : 2u * (q - nslabs) + 1u;
    const u32 n = meta[blk].n;
    if (n == 0u) continue;
    const size_t off = lbz_elem_off(L, blk);
    bwt_block(Tbase + off, n, Bbase + off, &meta[blk], s, &S);
  }
}
