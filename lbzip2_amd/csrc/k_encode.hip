/*
 * k_encode.hip -- stages 4+5: prefix-code selection (8 rounds of EM over <=6 tables,
 * length-limited codes by package-merge, table renumbering) and MSB-first bit packing of
 * one block, one block per workgroup, every table resident in LDS.
 *
 * Replaces generate_prefix_code() and helpers (reference src/encode.c:553-1137), the size
 * and padding logic of encode() (encode.c:460-545) and transmit() (encode.c:1152-1281).
 * The choices the reference makes are not the unique optimum, so its arithmetic is
 * followed exactly where it decides ties:
 *   - weights  freq<<32 | depth<<24 | count<<16 | (258-symbol)   (encode.c:732-741, 900-902)
 *   - E-step cost sums in six packed 10-bit fields that may carry (encode.c:1050-1061, 858-872)
 *   - two-queue Huffman merge with its </<= asymmetry             (encode.c:574-615)
 *   - height search 2..20 keeping the first strict minimum        (encode.c:913-945)
 * What is parallel here: the E-step (one lane per 50-symbol group, histograms by LDS
 * atomics), every sort (rank by counting), the package-merge lists (each level is one
 * parallel merge of leaves and packages by binary search), the per-height cost evaluation,
 * and the bit packer (code lengths -> workgroup add-scan -> each lane ORs its codes into an
 * LDS window that is flushed as big-endian words), the selector MTF (a max-scan per table).  What stays serial
 * on one lane: the 257-step Huffman merge per table (a wave per table; the rest of the M-step is the wave's).
 * 512 threads and 73 KB of LDS: two blocks share a CU (enc_lds below).
 *
 * Traffic: 9 passes over the MTF symbols (8 E-steps + packing) = 18 B per symbol + output.
 */
#include "lbz_common.h"
#undef LBZ_WG
#define LBZ_WG LBZ_ENCODE_WG
#undef LBZ_NW
#define LBZ_NW (LBZ_WG / 64)
#include "lbz_kernels.h"

#define PK_IPT 4u
#define PK_TILE (LBZ_WG * PK_IPT)
#define WIN_WORDS (PK_TILE * LBZ_MAX_CODELEN / 32u + 8u)
#define PM_LEVELS 20u
#define PM_ITEMS (2u * LBZ_MAX_ALPHA)
static_assert(LBZ_WG >= 320 && LBZ_WG % 64 == 0, "tables are filled one symbol per thread");

/* Three stretches of a block's coding need three sets of tables, one after the other: the EM rounds (cost fields, sorted
 * weights, the merge's tree), the length-limited codes (package-merge lists, one table at a time) and the packing (bit
 * window, selector bytes).  They share their LDS; what lives through all of them stands in front.  74 KB.            */
struct enc_lds {
  wg_scratch sc;
  u32 mfreq[LBZ_MAX_ALPHA + 2];
  u32 freq[LBZ_MAX_TREES][LBZ_MAX_ALPHA + 2];
  u8 len[LBZ_MAX_TREES][LBZ_MAX_ALPHA + 2];
  u32 lc[LBZ_MAX_TREES][LBZ_MAX_ALPHA + 2];      /* code << 5 | length (during the EM rounds: node depths of the M-step) */
  u8 sel[LBZ_MAX_SEL + 6];
  u32 firstpos[LBZ_MAX_TREES];
  u32 old2new[LBZ_MAX_TREES], new2old[LBZ_MAX_TREES];
  u8 inuse[256];
  u32 bc[8];
  union {
    struct {
      u64 pack[LBZ_MAX_ALPHA + 2];
      u64 wsort[LBZ_MAX_TREES][LBZ_MAX_ALPHA];
      u32 parent[LBZ_MAX_TREES][LBZ_MAX_ALPHA];
      u32 dcnt[LBZ_MAX_TREES][2][32];
    } em;
    struct {
      u64 wq[LBZ_MAX_ALPHA], wtmp[LBZ_MAX_ALPHA];
      u32 lfreq[LBZ_MAX_ALPHA];
      u32 items[2][PM_ITEMS];                    /* a level is built from the one below it only: two buffers */
      u16 leaves_in[PM_LEVELS + 1][PM_ITEMS + 2];
      u32 nitems[PM_LEVELS + 2];
      u32 taken[PM_LEVELS + 2][PM_LEVELS + 2];
      u16 upto[PM_LEVELS + 2][PM_LEVELS + 2];
      u8 hl[PM_LEVELS + 1][LBZ_MAX_ALPHA + 2];
      u32 hcost[PM_LEVELS + 2];
    } lim;
    struct {
      u32 win[WIN_WORDS];
      u8 selmtf[LBZ_MAX_SEL + 8];
    } pk;
  } u;
};

__device__ __forceinline__ u64 leaf_weight(u32 f, u32 sym)
{
  return ((u64)f << 32) | 0x10000ull | (u64)(LBZ_MAX_ALPHA - sym);
}

/* ---- E-step seeding: split the alphabet into nt classes of similar mass (encode.c:779-841) */
__device__ void seed_tables(enc_lds *S, u32 as, u32 nm, u32 nt)
{
  u32 live = 0, a = 0;
  for (u32 v = 0; v < as; v++) live += S->mfreq[v] != 0u;
  if (nt > live) nt = live;
  for (u32 t = 0; nt > 0u; t++, nt--) {
    u32 f = S->mfreq[a], cum = f, b = a + 1u;
    live -= f != 0u;
    while (live > nt - 1u && cum * nt < nm) {
      f = S->mfreq[b++]; cum += f; live -= f != 0u;
    }
    if (cum > f && (2u * cum - f) * nt > 2u * nm) {
      cum -= f; live += f != 0u; b--;
    }
    for (u32 v = a; v < b; v++) S->len[t][v] = 0;
    a = b;
    nm -= cum;
  }
}

/* ---- M-step for one table on one WAVE: unrestricted Huffman lengths (encode.c:713-766).
 * The two-queue merge is a chain of as - 1 decisions and stays on lane 0, but the heads of both queues (two leaves, two
 * internal nodes) live in registers: a step compares registers and then refills them, one LDS round trip where a loop that
 * reads its operands as it needs them took five or six in a row (0.11 ms per table and EM round, 0.89 ms per block).
 * What follows the merge is the wave's: node depths by relaxation from the root (a node's parent has a smaller index; the
 * tree is ~20 deep), their histogram, the leaves' depths from the counts per level.                                   */
__device__ void huffman_lengths_wave(enc_lds *S, u32 t, u32 as)
{
  const u32 lane = threadIdx.x & 63u;
  u64 *w = S->u.em.wsort[t];
  u32 *par = S->u.em.parent[t];
  u32 *depth = S->lc[t];                                /* free until limited_code() writes the final codes */
  u32 *internal_at = S->u.em.dcnt[t][0], *leaves_at = S->u.em.dcnt[t][1];
  constexpr u32 UNK = 0xFFFFFFFFu;
  for (u32 i = lane; i < as; i += 64u) depth[i] = i == 1u ? 0u : UNK;
  if (lane < 32u) internal_at[lane] = 0;
  if (lane == 0u) {
    /* No branch inside a step: on one lane every instruction is four cycles of an otherwise idle SIMD, and a divergent
       branch costs a dozen of them in exec-mask bookkeeping.  Stores that a case does not make go to a dummy slot (index 0
       of par[], never read), loads are made for every case and selected. */
    u32 leaf = as, node = as;
    u64 L1 = w[as - 1u], L2 = as >= 2u ? w[as - 2u] : 0ull, N1 = 0, N2 = 0;
    u64 keep = w[as - 1u] & 0xFFFFull;                  /* slot x held a leaf that is merged by now: its symbol stays (read below) */
    for (u32 x = as - 1u; x > 0u; x--) {
      const u32 n_int = node - 1u - x;
      const bool two_nodes = leaf == 0u || (n_int >= 2u && N2 < L1);
      const bool two_leaves = !two_nodes && (n_int == 0u || (leaf >= 2u && L2 <= N1));
      const bool mixed = !two_nodes && !two_leaves;
      const u64 a = two_leaves ? L1 : N1;
      const u64 b = two_nodes ? N2 : (two_leaves ? L2 : L1);
      par[two_leaves ? 0u : node - 1u] = x;
      par[two_nodes ? node - 2u : 0u] = x;
      node -= two_nodes ? 2u : (mixed ? 1u : 0u);
      leaf -= two_leaves ? 2u : (mixed ? 1u : 0u);
      const u64 da = a & 0xFF000000ull, db = b & 0xFF000000ull;
      const u64 nv = keep + ((a + b) & ~0xFF00FFFFull) + (da > db ? da : db) + 0x01000000ull;
      w[x] = nv;
      /* the queues' heads for the next step: internal nodes x .. node-1 (oldest = lightest at node-1), leaves below `leaf` */
      const u64 wn1 = w[node - 1u], wn2 = w[node >= 2u ? node - 2u : 0u];
      const u64 wl1 = w[leaf >= 1u ? leaf - 1u : 0u], wl2 = w[leaf >= 2u ? leaf - 2u : 0u];
      keep = w[x - 1u] & 0xFFFFull;
      const u32 q = node - x;
      N1 = node - 1u == x ? nv : wn1;
      N2 = q >= 2u ? (node - 2u == x ? nv : wn2) : 0ull;
      L1 = leaf >= 1u ? wl1 : 0ull;
      L2 = leaf >= 2u ? wl2 : 0ull;
    }
  }
  wave_sync();
  /* depths of the internal nodes 2 .. as-1 (node 1 is the root) */
  u32 pj[5];
  u32 pend = 0;
#pragma unroll
  for (u32 j = 0; j < 5u; j++) {
    const u32 i = lane + 64u * j;
    pj[j] = 0;
    if (i >= 2u && i < as) { pj[j] = par[i]; pend |= 1u << j; }
  }
  while (__ballot(pend != 0u)) {
#pragma unroll
    for (u32 j = 0; j < 5u; j++)
      if (pend & (1u << j)) {
        const u32 dp = depth[pj[j]];
        if (dp != UNK) { depth[lane + 64u * j] = dp + 1u; pend &= ~(1u << j); }
      }
    wave_sync();
  }
#pragma unroll
  for (u32 j = 0; j < 5u; j++) {
    const u32 i = lane + 64u * j;
    if (i >= 1u && i < as) atomicAdd(&internal_at[depth[i]], 1u);
  }
  wave_sync();
  /* leaves per level, then their running total: leaf number i (heaviest first) sits on the first level whose total passes i */
  u32 la = 0;
  if (lane >= 1u && lane <= 30u) la = 2u * internal_at[lane - 1u] - internal_at[lane];
  const u32 cum = wave_incl_add(la);
  wave_sync();
  if (lane < 32u) leaves_at[lane] = cum;
  wave_sync();
#pragma unroll
  for (u32 j = 0; j < 5u; j++) {
    const u32 i = lane + 64u * j;
    if (i < as) {
      u32 d = 1u;
      while (d < 30u && leaves_at[d] <= i) d++;
      S->len[t][LBZ_MAX_ALPHA - (u32)(w[i] & 0xFFFFull)] = (u8)d;
    }
  }
}

/* ---- descending sort of as weights by counting, whole workgroup: dst[rank] = src weight */
__device__ __forceinline__ u32 rank_desc(const u64 *w, u32 as, u32 i)
{
  const u64 me = w[i];
  u32 r = 0, j = 0;
  for (; j + 8u <= as; j += 8u) {                       /* eight reads in flight: one by one the loop is a chain of LDS round trips */
    u64 v[8];
#pragma unroll
    for (u32 k = 0; k < 8u; k++) v[k] = w[j + k];
#pragma unroll
    for (u32 k = 0; k < 8u; k++) r += v[k] > me;
  }
  for (; j < as; j++) r += w[j] > me;
  return r;
}

/* ---- length-limited code of table t by package-merge (encode.c:660-710, 882-987);
 *      all threads call; returns the bit cost of the table and its symbols.            */
__device__ u32 limited_code(enc_lds *S, u32 t, u32 as)
{
  const u32 tid = threadIdx.x;
  u64 *wq = S->u.lim.wq;
  u64 *wtmp = S->u.lim.wtmp;
  const u32 want = 2u * as - 2u;

  if (tid < as) wtmp[tid] = leaf_weight(S->freq[t][tid], tid);
  __syncthreads();
  if (tid < as) wq[rank_desc(wtmp, as, tid)] = wtmp[tid];
  __syncthreads();
  if (tid < as) {
    const u32 f = (u32)(wq[as - 1u - tid] >> 32);       /* ascending */
    S->u.lim.lfreq[tid] = f;
    S->u.lim.items[1][tid] = f;
    S->u.lim.leaves_in[1][tid + 1u] = (u16)(tid + 1u);
  }
  if (tid == 0) { S->u.lim.nitems[1] = as; S->u.lim.leaves_in[1][0] = 0; }
  __syncthreads();

  /* one barrier per level: the item counts follow from as alone (every thread keeps them), a package's weight is the
     sum of two items of the level below and is formed where it is compared */
  u32 nprev = as;
  for (u32 lv = 2; lv <= PM_LEVELS; lv++) {
    const u32 npk = nprev / 2u;
    const u32 *below = S->u.lim.items[(lv - 1u) & 1u];
    for (u32 e = tid; e < as + npk; e += LBZ_WG) {
      if (e < as) {                                     /* leaf: packages strictly lighter go first */
        const u32 f = S->u.lim.lfreq[e];
        u32 lo = 0, hi = npk;
        while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (below[2u * mid] + below[2u * mid + 1u] < f) lo = mid + 1u; else hi = mid; }
        const u32 pos = e + lo;
        if (pos < want) { S->u.lim.items[lv & 1u][pos] = f; S->u.lim.leaves_in[lv][pos + 1u] = (u16)(e + 1u); }
      } else {                                          /* package: leaves of equal weight go first */
        const u32 k = e - as;
        const u32 f = below[2u * k] + below[2u * k + 1u];
        u32 lo = 0, hi = as;
        while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (S->u.lim.lfreq[mid] <= f) lo = mid + 1u; else hi = mid; }
        const u32 pos = k + lo;
        if (pos < want) { S->u.lim.items[lv & 1u][pos] = f; S->u.lim.leaves_in[lv][pos + 1u] = (u16)lo; }
      }
    }
    const u32 tot = as + npk;
    nprev = tot < want ? tot : want;
    if (tid == 0) { S->u.lim.nitems[lv] = nprev; S->u.lim.leaves_in[lv][0] = 0; }
    __syncthreads();
  }

  /* taken[h][d] = leaves used on level h-d under height limit h; upto[h][d] = symbols (heaviest first) that the reference's
     loop over the levels has given a length <= d -- kept exactly as that loop counts, saturation at `as` included */
  if (tid >= 1u && tid <= PM_LEVELS) {
    const u32 h = tid;
    u32 k = want < S->u.lim.nitems[h] ? want : S->u.lim.nitems[h];
    for (u32 d = 0; d <= PM_LEVELS; d++) S->u.lim.taken[h][d] = 0;
    for (u32 d = 0; d < h; d++) {
      const u32 lv = h - d;
      if (k > S->u.lim.nitems[lv]) k = S->u.lim.nitems[lv];
      const u32 nl = S->u.lim.leaves_in[lv][k];
      S->u.lim.taken[h][d] = nl;
      k = 2u * (k - nl);
    }
    u32 cum = 0;
    S->u.lim.upto[h][0] = 0;
    for (u32 d = 1; d <= h; d++) {
      const u64 c = (u64)cum + (u64)(u32)(S->u.lim.taken[h][d - 1u] - S->u.lim.taken[h][d]);
      cum = c < (u64)as ? (u32)c : as;
      S->u.lim.upto[h][d] = (u16)cum;
    }
    S->u.lim.hcost[h] = (h >= 2u && (1u << h) >= as) ? 5u + as : 0u;
  }
  __syncthreads();

  /* cost of every height limit (encode.c:922-938), a (limit, symbol) pair per thread */
  for (u32 e = tid; e < (PM_LEVELS - 1u) * as; e += LBZ_WG) {
    const u32 h = 2u + e / as, rank = e % as;
    if ((1u << h) < as) continue;
    u32 d = 1u;
    while (d <= h && (u32)S->u.lim.upto[h][d] <= rank) d++;
    if (d <= h) {
      const u64 wr = wq[rank];
      S->u.lim.hl[h][LBZ_MAX_ALPHA - (u32)(wr & 0xFFFFull)] = (u8)d;
      atomicAdd(&S->u.lim.hcost[h], (u32)(wr >> 32) * d);
    }
  }
  __syncthreads();
  for (u32 e = tid; e < (PM_LEVELS - 1u) * as; e += LBZ_WG) {
    const u32 h = 2u + e / as, v = e % as;
    if ((1u << h) < as || v == 0u) continue;
    const int dl = (int)S->u.lim.hl[h][v] - (int)S->u.lim.hl[h][v - 1u];
    if (dl) atomicAdd(&S->u.lim.hcost[h], 2u * (u32)(dl < 0 ? -dl : dl));
  }
  __syncthreads();
  if (tid == 0) {                                        /* first strict minimum, encode.c:913-945 */
    u32 best_cost = 0xFFFFFFFFu, best_h = PM_LEVELS;
    for (u32 h = 2; h <= PM_LEVELS; h++) {
      if ((1u << h) < as) continue;
      if (S->u.lim.taken[h][h - 1u] == 0u) break;
      if (S->u.lim.hcost[h] < best_cost) { best_cost = S->u.lim.hcost[h]; best_h = h; }
    }
    S->bc[0] = best_cost;
    S->bc[1] = best_h;
    u32 next = 0;                                        /* canonical first codes per length */
    for (u32 d = 1; d <= best_h; d++) {
      const u32 k = S->u.lim.taken[best_h][d - 1u] - S->u.lim.taken[best_h][d];
      S->u.lim.hcost[d] = next;
      next = (next + k) << 1;
    }
  }
  __syncthreads();
  const u32 best_cost = S->bc[0], best_h = S->bc[1];
  if (tid < as) {
    const u32 l = S->u.lim.hl[best_h][tid];
    u32 same = 0, v = 0;
    for (; v + 8u <= tid; v += 8u) {
      u32 x[8];
#pragma unroll
      for (u32 k = 0; k < 8u; k++) x[k] = S->u.lim.hl[best_h][v + k];
#pragma unroll
      for (u32 k = 0; k < 8u; k++) same += x[k] == l;
    }
    for (; v < tid; v++) same += S->u.lim.hl[best_h][v] == l;
    S->len[t][tid] = (u8)l;
    S->lc[t][tid] = ((S->u.lim.hcost[l] + same) << 5) | l;
  }
  if (tid == as) { S->len[t][as] = 0; S->lc[t][as] = 0; }
  __syncthreads();
  return best_cost;
}

/* ---- bit sink: MSB-first deposit into an LDS window of native-order words ------------ */
__device__ __forceinline__ void put_bits(u32 *win, u32 wbase, u64 bitpos, u32 nbits, u32 val)
{
  if (nbits == 0u || val == 0u) return;
  const u32 wi = (u32)(bitpos >> 5) - wbase;
  const u32 off = (u32)bitpos & 31u;
  const u64 v = (u64)val << (64u - off - nbits);
  const u32 hi = (u32)(v >> 32), lo = (u32)v;
  if (hi) atomicOr(&win[wi], hi);
  if (lo) atomicOr(&win[wi + 1u], lo);
}

__device__ __forceinline__ u32 bswap32(u32 x)
{
  return (x << 24) | ((x & 0xFF00u) << 8) | ((x >> 8) & 0xFF00u) | (x >> 24);
}

/* Write out every complete word below endbit, keep the partial one.  All threads call. */
__device__ u32 flush_window(u32 *win, u32 *out32, u32 wbase, u64 endbit)
{
  const u32 tid = threadIdx.x;
  __syncthreads();
  const u32 complete = (u32)(endbit >> 5) - wbase;
  const u32 carry = win[complete];
  for (u32 i = tid; i < complete; i += LBZ_WG) out32[wbase + i] = bswap32(win[i]);
  __syncthreads();
  for (u32 i = tid; i <= complete + 1u; i += LBZ_WG) win[i] = (i == 0u) ? carry : 0u;
  __syncthreads();
  return wbase + complete;
}

__global__ void __launch_bounds__(LBZ_WG, 4)
k_encode(const u16 *Vbase, const u32 *freq_in, u8 *Obase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, const u32 *slabs)
{
  __shared__ enc_lds S;
  const u32 tid = threadIdx.x;
  const u32 blk = lbz_round_block(first, count, blockIdx.x, slabs);
  lbz_block_meta *M = &meta[blk];
  if (M->n == 0u) return;
  const u16 *mtfv = Vbase + lbz_elem_off(L, blk);
  const u32 *mtfv32 = reinterpret_cast<const u32 *>(mtfv);
  u32 *out32 = reinterpret_cast<u32 *>(Obase + lbz_out_off(L, blk));
  const u32 out_cap = (blk & 1u) ? L.out_b : L.out_a;

  const u32 nm = M->nmtf, as = M->alpha;
  const u32 ns = (nm + LBZ_GROUP - 1u) / LBZ_GROUP;
  const u32 nt = nm > 2400u ? 6u : nm > 1200u ? 5u : nm > 600u ? 4u : nm > 300u ? 3u : nm > 150u ? 2u : 1u;

  for (u32 i = tid; i < LBZ_MAX_ALPHA + 2u; i += LBZ_WG) S.mfreq[i] = i < as ? freq_in[(size_t)blk * 260u + i] : 0u;
  for (u32 i = tid; i < LBZ_MAX_TREES * (LBZ_MAX_ALPHA + 2u); i += LBZ_WG) (&S.len[0][0])[i] = 1;
  if (tid < 256u) S.inuse[tid] = M->inuse[tid];
  __syncthreads();
  if (tid == 0) seed_tables(&S, as, nm, nt);
  __syncthreads();

#ifdef ENC_TICKS
  const u64 tk0 = wall_clock64();
  u64 tke = 0, tkm = 0, tks = 0;
#endif
  /* ---- EM (encode.c:1043-1084) ---- */
  for (u32 it = 0; it < LBZ_CLUSTER; it++) {
#ifdef ENC_TICKS
    const u64 ta = wall_clock64();
#endif
    if (tid <= as) {
      u64 x = 0;
      if (tid < as)
        for (int t = (int)LBZ_MAX_TREES - 1; t >= 0; t--) x = (x << 10) + S.len[t][tid];
      S.u.em.pack[tid] = x;
    }
    for (u32 i = tid; i < LBZ_MAX_TREES * (LBZ_MAX_ALPHA + 2u); i += LBZ_WG) (&S.freq[0][0])[i] = 0;
    __syncthreads();

    for (u32 g = tid; g < ns; g += LBZ_WG) {
      u32 sy[LBZ_GROUP / 2u];
      u64 sum = 0;
#pragma unroll
      for (u32 i = 0; i < LBZ_GROUP / 2u; i++) {
        sy[i] = mtfv32[g * (LBZ_GROUP / 2u) + i];
        sum += S.u.em.pack[sy[i] & 0xFFFFu] + S.u.em.pack[sy[i] >> 16];
      }
      u32 bt = 0, bcst = (u32)(sum & 0x3FFull);
      for (u32 t = 1; t < nt; t++) {
        sum >>= 10;
        const u32 c = (u32)(sum & 0x3FFull);
        if (c < bcst) { bcst = c; bt = t; }
      }
      S.sel[g] = (u8)bt;
#pragma unroll
      for (u32 i = 0; i < LBZ_GROUP / 2u; i++) {
        atomicAdd(&S.freq[bt][sy[i] & 0xFFFFu], 1u);
        atomicAdd(&S.freq[bt][sy[i] >> 16], 1u);
      }
    }
    __syncthreads();

#ifdef ENC_TICKS
    const u64 tb = wall_clock64();
    tke += tb - ta;
#endif
    /* M-step: sort each table's weights (all threads), then one lane per table merges */
    for (u32 e = tid; e < nt * as; e += LBZ_WG) {
      const u32 t = e / as, i = e - t * as;
      const u32 f = S.freq[t][i];
      S.u.em.wsort[t][i] = leaf_weight(f ? f : 1u, i);
    }
    __syncthreads();
    constexpr u32 MW = (LBZ_MAX_TREES * LBZ_MAX_ALPHA + LBZ_WG - 1u) / LBZ_WG;   /* weights per thread */
    u64 mine[MW]; u32 rk2[MW]; u32 cntm = 0;
    for (u32 e = tid; e < nt * as; e += LBZ_WG) {
      const u32 t = e / as, i = e - t * as;
      mine[cntm] = S.u.em.wsort[t][i];
      rk2[cntm] = rank_desc(S.u.em.wsort[t], as, i);
      cntm++;
    }
    __syncthreads();
    cntm = 0;
    for (u32 e = tid; e < nt * as; e += LBZ_WG) {
      const u32 t = e / as;
      S.u.em.wsort[t][rk2[cntm]] = mine[cntm];
      cntm++;
    }
    __syncthreads();
#ifdef ENC_TICKS
    tks += wall_clock64() - tb;
#endif
    if ((tid >> 6) < nt) huffman_lengths_wave(&S, tid >> 6, as);
    __syncthreads();
#ifdef ENC_TICKS
    tkm += wall_clock64() - tb;
#endif
  }
#ifdef ENC_TICKS
  const u64 tk1 = wall_clock64();
#endif

  /* ---- renumber tables by first use (encode.c:1088-1111) ---- */
  if (tid < LBZ_MAX_TREES) S.firstpos[tid] = 0xFFFFFFFFu;
  __syncthreads();
  for (u32 g = tid; g < ns; g += LBZ_WG) atomicMin(&S.firstpos[S.sel[g]], g);
  __syncthreads();
  if (tid == 0) {
    u32 used = 0;
    for (;;) {
      u32 bt = 0xFFFFFFFFu, bp = 0xFFFFFFFFu;
      for (u32 t = 0; t < nt; t++)
        if (S.firstpos[t] < bp) { bp = S.firstpos[t]; bt = t; }
      if (bt == 0xFFFFFFFFu) break;
      S.firstpos[bt] = 0xFFFFFFFFu;
      S.old2new[bt] = used;
      S.new2old[used] = bt;
      used++;
    }
    S.bc[4] = used;
  }
  __syncthreads();
  u32 used = S.bc[4];
  u32 cost = 0;
  for (u32 u = 0; u < used; u++) cost += limited_code(&S, S.new2old[u], as);

  if (used == 1u) {                                      /* dummy second table, encode.c:1117-1132 */
    const u32 t = S.new2old[0] ^ 1u;
    u32 lg = 0;
    while ((2u << lg) <= as) lg++;
    const u32 nshort = (2u << lg) - as;
    if (tid < as) S.len[t][tid] = (u8)(tid < nshort ? lg : lg + 1u);
    if (tid == 0) { S.old2new[t] = 1; S.new2old[1] = t; }
    if (nshort < as) cost += 2u;
    cost += as + 5u;
    used = 2u;
    __syncthreads();
  }
  /* the packing's tables take the place of the code tables from here on (limited_code() ends on a barrier; the scans of the
     selector stage come before the first bit is deposited) */
  for (u32 i = tid; i < WIN_WORDS; i += LBZ_WG) S.u.pk.win[i] = 0;

#ifdef ENC_TICKS
  const u64 tk2 = wall_clock64();
#endif
  /* ---- selector MTF, exact size, padding (encode.c:473-545) ---- */
  u8 *selmtf = S.u.pk.selmtf;
  /* Selector MTF (encode.c:473-492).  As in k_mtf: the rank of a group's table is the number of
     tables used more recently, so all that is serial is "where was table t last used before
     group g" -- one exclusive max-scan per table.  Positions are kept as g + 7; a table not used
     yet sits at 6 - t, i.e. in the initial order 0,1,..,5.                                   */
  u32 selbits;
  {
    constexpr u32 SEL_IPT = 8u;
    u32 carry[LBZ_MAX_TREES], mybits = 0;
#pragma unroll
    for (u32 t = 0; t < LBZ_MAX_TREES; t++) carry[t] = 6u - t;
    for (u32 t0 = 0; t0 < ns; t0 += LBZ_WG * SEL_IPT) {
      const u32 g0 = t0 + tid * SEL_IPT;
      u32 cc[SEL_IPT], mylast[LBZ_MAX_TREES];
#pragma unroll
      for (u32 t = 0; t < LBZ_MAX_TREES; t++) mylast[t] = 0;
#pragma unroll
      for (u32 k = 0; k < SEL_IPT; k++) {
        const u32 g = g0 + k;
        cc[k] = g < ns ? S.old2new[S.sel[g]] : 0xFFu;
#pragma unroll
        for (u32 t = 0; t < LBZ_MAX_TREES; t++) if (cc[k] == t) mylast[t] = g + 7u;
      }
      u32 last[LBZ_MAX_TREES];
#pragma unroll
      for (u32 t = 0; t < LBZ_MAX_TREES; t++) {
        u32 e, d0, tm, d1;
        wg_excl_max_add(mylast[t], 0u, &e, &d0, &tm, &d1, &S.sc);
        last[t] = e > carry[t] ? e : carry[t];
        carry[t] = tm > carry[t] ? tm : carry[t];
      }
#pragma unroll
      for (u32 k = 0; k < SEL_IPT; k++) {
        const u32 g = g0 + k;
        if (g < ns) {
          u32 lc = 0;
#pragma unroll
          for (u32 t = 0; t < LBZ_MAX_TREES; t++) lc = (cc[k] == t) ? last[t] : lc;
          u32 j = 0;
#pragma unroll
          for (u32 t = 0; t < LBZ_MAX_TREES; t++) j += last[t] > lc ? 1u : 0u;
          selmtf[g] = (u8)j;
          mybits += j + 1u;
#pragma unroll
          for (u32 t = 0; t < LBZ_MAX_TREES; t++) if (cc[k] == t) last[t] = g + 7u;
        }
      }
    }
    selbits = wg_sum(mybits, &S.sc);
  }
  if (tid == 0) {
    u32 bits = 48u + 32u + 1u + 24u + 3u + 15u + cost + selbits;
    const u32 padbits = (8u - (bits & 7u)) & 7u;
    bits += padbits;
    u32 nstx = ns;
    if (padbits & 1u) selmtf[nstx++] = 0;
    bits += 16u;
    u32 big = 0;
    for (u32 i = 0; i < 16u; i++) {
      u32 any = 0;
      for (u32 j = 0; j < 16u; j++) any |= S.inuse[16u * i + j];
      if (any) { bits += 16u; big |= 0x8000u >> i; }
    }
    S.bc[0] = bits >> 3;
    S.bc[1] = padbits >> 1;          /* tree_pad */
    S.bc[2] = nstx;
    S.bc[3] = big;
  }
  __syncthreads();
  const u32 out_len = S.bc[0], tree_pad = S.bc[1], nstx = S.bc[2], big = S.bc[3];
  if (out_len + 8u > out_cap) {      /* cannot happen for sane inputs; flag instead of overrun */
    if (tid == 0) { M->err = 2u; M->out_len = 0; }
    return;
  }

#ifdef ENC_TICKS
  const u64 tk3 = wall_clock64();
#endif
  /* ---- packing (encode.c:1185-1278) ---- */
  u32 wbase = 0;
  u64 bitpos = 0;
  if (tid == 0) {
    u64 bp = 0;
    put_bits(S.u.pk.win, 0, bp, 24, 0x314159u); bp += 24;
    put_bits(S.u.pk.win, 0, bp, 24, 0x265359u); bp += 24;
    put_bits(S.u.pk.win, 0, bp, 32, ~M->crc); bp += 32;
    bp += 1;                                              /* not randomised */
    put_bits(S.u.pk.win, 0, bp, 24, M->bwt_idx); bp += 24;
    put_bits(S.u.pk.win, 0, bp, 16, big); bp += 16;
    for (u32 i = 0; i < 16u; i++)
      if (big & (0x8000u >> i)) {
        u32 pk = 0;
        for (u32 j = 0; j < 16u; j++) pk = (pk << 1) | (S.inuse[16u * i + j] ? 1u : 0u);
        put_bits(S.u.pk.win, 0, bp, 16, pk); bp += 16;
      }
    put_bits(S.u.pk.win, 0, bp, 3, used); bp += 3;
    put_bits(S.u.pk.win, 0, bp, 15, nstx); bp += 15;
    S.bc[5] = (u32)bp;
  }
  __syncthreads();
  bitpos = S.bc[5];
  wbase = flush_window(S.u.pk.win, out32, wbase, bitpos);

  /* selectors: value j as j ones and a zero */
  for (u32 t0 = 0; t0 < nstx; t0 += PK_TILE) {
    const u32 i0 = t0 + tid * PK_IPT;
    u32 nb = 0;
#pragma unroll
    for (u32 k = 0; k < PK_IPT; k++) if (i0 + k < nstx) nb += selmtf[i0 + k] + 1u;
    u32 tot;
    u64 bp = bitpos + wg_excl_add(nb, &tot, &S.sc);
#pragma unroll
    for (u32 k = 0; k < PK_IPT; k++)
      if (i0 + k < nstx) {
        const u32 v = selmtf[i0 + k] + 1u;
        put_bits(S.u.pk.win, wbase, bp, v, (1u << v) - 2u);
        bp += v;
      }
    bitpos += tot;
    wbase = flush_window(S.u.pk.win, out32, wbase, bitpos);
  }

  /* code-length tables: 5-bit start, then +/-1 steps "10"/"11" and a "0" per symbol */
  {
    const u32 per = as + 1u, nitem = used * per;
    for (u32 t0 = 0; t0 < nitem; t0 += LBZ_WG) {
      const u32 e = t0 + tid;
      u32 nb = 0, up = 0, steps = 0, first5 = 0;
      bool is_first = false;
      if (e < nitem) {
        const u32 u = e / per, k = e - u * per;
        const u8 *len = S.len[S.new2old[u]];
        int a0 = len[0];
        if (u == 0u) a0 += (a0 < 4) ? (int)tree_pad : -(int)tree_pad;
        if (k == 0u) { is_first = true; first5 = (u32)a0; nb = 5u; }
        else {
          const int prev = (k == 1u) ? a0 : (int)len[k - 2u];
          const int cur = (int)len[k - 1u];
          up = cur > prev;
          steps = (u32)(up ? cur - prev : prev - cur);
          nb = 2u * steps + 1u;
        }
      }
      u32 tot;
      u64 bp = bitpos + wg_excl_add(nb, &tot, &S.sc);
      if (e < nitem) {
        if (is_first) put_bits(S.u.pk.win, wbase, bp, 5, first5);
        else {
          u32 left = steps;
          while (left) {                                 /* <=16 steps (32 bits) per deposit */
            const u32 c = left < 16u ? left : 16u;
            const u32 pat = up ? 0xAAAAAAAAu : 0xFFFFFFFFu;
            put_bits(S.u.pk.win, wbase, bp, 2u * c, pat >> (32u - 2u * c));
            bp += 2u * c;
            left -= c;
          }
          /* terminating 0 bit: nothing to OR */
        }
      }
      bitpos += tot;
      wbase = flush_window(S.u.pk.win, out32, wbase, bitpos);
    }
  }

  /* symbols: code of each of the group's 50 values from the group's table */
  {
    const u32 nsym = ns * LBZ_GROUP;
    for (u32 t0 = 0; t0 < nsym; t0 += PK_TILE) {
      const u32 i0 = t0 + tid * PK_IPT;
      u32 lcv[PK_IPT];
      u32 nb = 0;
      if (i0 < nsym) {                                   /* nsym and i0 are multiples of 2 */
        const u32 a = mtfv32[i0 >> 1];
        const u32 b = (i0 + 2u < nsym) ? mtfv32[(i0 >> 1) + 1u] : 0u;
        const u32 sy[PK_IPT] = { a & 0xFFFFu, a >> 16, b & 0xFFFFu, b >> 16 };
#pragma unroll
        for (u32 k = 0; k < PK_IPT; k++) {
          const u32 i = i0 + k;
          lcv[k] = (i < nsym) ? S.lc[S.sel[i / LBZ_GROUP]][sy[k]] : 0u;
          nb += lcv[k] & 31u;
        }
      } else {
#pragma unroll
        for (u32 k = 0; k < PK_IPT; k++) lcv[k] = 0;
      }
      u32 tot;
      u64 bp = bitpos + wg_excl_add(nb, &tot, &S.sc);
#pragma unroll
      for (u32 k = 0; k < PK_IPT; k++) {
        const u32 l = lcv[k] & 31u;
        put_bits(S.u.pk.win, wbase, bp, l, lcv[k] >> 5);
        bp += l;
      }
      bitpos += tot;
      wbase = flush_window(S.u.pk.win, out32, wbase, bitpos);
    }
  }
  if (tid == 0) {
    if (bitpos & 31ull) out32[wbase] = bswap32(S.u.pk.win[0]);
    M->out_len = out_len;
    M->num_trees = used;
    M->num_sel = nstx;
    if ((u32)(bitpos >> 3) != out_len || (bitpos & 7ull)) M->err = 3u;   /* cf. encode.c:1275-1277 */
#ifdef ENC_TICKS
    M->ticks[1] = (u32)tke; M->ticks[2] = (u32)tkm; M->ticks[3] = (u32)(tk2 - tk1); M->ticks[4] = (u32)(tk3 - tk2);
    M->ticks[5] = (u32)(wall_clock64() - tk3); M->ticks[6] = (u32)(tk0 & 0xffffffffu); M->ticks[7] = (u32)tks;
#endif
  }
}
