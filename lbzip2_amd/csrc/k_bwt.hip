/*
 * k_bwt.hip -- stage 2: Burrows-Wheeler transform of the CYCLIC rotations of a block.
 *
 * Replaces divbwt() (reference src/divbwt.c:1706-1726; its sort_typeBstar/sssort/trsort/construct_BWT machinery,
 * divbwt.c:1488-1699, is a serial induced-sorting design with no data-parallel analogue).  The BWT byte string is
 * mathematically unique, so any correct rotation sorter reproduces it.  This one is a most-significant-digit string sort:
 *
 *   keys         rotation i gets a 64-bit key: the dense codes (b = ceil(log2 #used bytes) bits each; the bytes themselves
 *                when more than 128 values are in use) of its first S = 64/b symbols, and a 32-bit value
 *                (code of the preceding byte << 24 | i): the BWT output byte rides along with the index.
 *   k_bwt_hist, k_bwt_scat x 4, k_bwt_segs   the partition: stable 8-bit radix passes in HBM on the key's top 32 bits (16
 *                for a block whose byte histogram is flat), one launch per pass, a block's rows dealt over up to 16
 *                workgroups (ranges of the pass's input; a pass counts the next pass's digits per range while it
 *                scatters).  Keys are built on the fly from the block text streamed through an LDS tile (the first pass
 *                never reads a key array); per-wave digit counters in LDS, ranks inside a wave from 8 ballots per row, a
 *                tile regrouped by digit in LDS so that every write is a run of equal-digit rows; the next tile's rows are
 *                in flight while a tile is ranked (unconditional global loads: lbz_asm.h, ldg_*).  k_bwt_segs fixes the
 *                SEGMENTS: 32 cuts of the sorted rows at group boundaries -- from here on every (block, segment)
 *                is a workgroup.
 *   k_bwt_part   the same partition as one workgroup per block, every pass in one launch: what launch_sort picks when big
 *                rounds run side by side on several streams.
 *   k_bwt_batch  consecutive whole groups of <= 1024 rows are pulled into LDS.  Waves claim chunks of groups (longest
 *                first); rows of a short group are placed by counting smaller keys, long groups are radix-sorted by their
 *                wave.  A finished batch writes 1 B (BWT byte) + 4 B (suffix-array entry) per rotation; rows whose 64-bit
 *                keys tie (two thirds of a text block) go, run by run, to the segment's LIST: (suffix, rank = first row of
 *                the run, symbols shared).  A group larger than a batch is sorted on its remaining key bits by the HBM
 *                radix sorter first; more than a batch of EQUAL keys joins the list as one long run.
 *   k_bwt_deep   the text rounds, LBZ_DEEP_ROUNDS launches: the listed runs are ordered by the text itself (0.9 MB per
 *                block, read-only) -- runs of up to 63 rows a 64-lane strip at a time in registers (all-pairs counting on
 *                52-bit slices of the next 16 bytes), longer runs taken apart symbol by symbol by a counting split.  From
 *                the third launch on the rows still tied (long repeats) have rank entries and step by ranks (prefix
 *                doubling over a SPARSE rank table: only tied rows have entries, a bit map says which).
 *   k_bwt_fix0, k_bwt_fixr x 17, k_bwt_fixend   the rank rounds, the fall-back: blocks the text rounds are not made for
 *                (mostly long runs, or two fifths of the rows still tied after the first launch: source trees, tables,
 *                periodic data) or did not finish.  Full prefix doubling: a rank for EVERY rotation (isa[], 8-byte
 *                {rank, rank before, tag} entries so that a launch reads the ranks as they stood when it began), one
 *                launch per doubling depth starting at the depth the block's ties are known to share (deep_h0).  Rows
 *                tied at depth >= n belong to an exactly periodic block (T = u^k): the origin pointer is the smallest
 *                equal row (the reference's choice among the k equal rows is an artefact of its unstable quicksort,
 *                SURVEY.md 8a-4 -- documented divergence, identical BWT bytes).
 *
 * Algorithmic traffic of the stage as priced in SURVEY.md 8(d): read T (1) + write SA (4) + read SA (4) + gather T (1) +
 * write BWT (1) = 11 B per block byte; measured HBM traffic is in profiles/README.md.  ticks[] / fticks[] in the block
 * record are diagnostics (tests/tools/quickperf.py, diag_deep.py).
 */
/* The kernels' geometry is their own constant: 256 threads (four workgroups per CU, 1024-row batches). */
#include "lbz_common.h"
#undef LBZ_WG
#define LBZ_WG LBZ_BWT_WG
#undef LBZ_NW
#define LBZ_NW (LBZ_WG / 64)
#include "lbz_kernels.h"

#define SORT_IPT 4u
#define SORT_TILE (LBZ_WG * SORT_IPT)
#define RANK_BITS 20u                   /* n <= 900000 < 2^20 */
#ifndef MSD_BITS
#define MSD_BITS 32u                    /* 8-bit partition passes in HBM: 16 = two, 24 = three, 32 = four.  With 1024-row batches a fourth
                                           pass pays: groups of more than a batch (HBM sorter) and of more than 256 rows (one wave's
                                           job) become rare: k_bwt_part +6.5 ms, k_bwt_batch -11 ms per 10^9 bytes of text */
#endif
/* ... per block: a block whose two leading key bytes are spread evenly (incompressible data: no byte value holds more
   than 1/128 of the rotations in either position) is partitioned on 16 bits only -- its groups are a dozen rows then, and
   two HBM passes over 12-byte rows are saved (k_bwt_part on random bytes: 46 -> 24 ms per 10^9 bytes).  The depth is in
   the block record (msd_bits) and, as a shift, in every sorting kernel's LDS header (msd_shift).                      */
#define MSD_BITS_FLAT 16u
#define PART_HALO 48u
#define PART_MAX 16u                    /* workgroups a block's rows are dealt over in the partition kernels (ranges of its input order) */
#define BATCH_CAP (LBZ_WG * 4u)          /* the tile a batch is loaded and (rarely) radix-sorted as: four rows per thread */
#ifndef BATCH_ROWS
#define BATCH_ROWS 832u                 /* rows a batch holds at most (<= BATCH_CAP): what the arrays of batch_lds are sized for.  Every phase of
                                           k_bwt_batch is a chain of LDS round trips, so what it needs is waves to hide them behind: 832 rows
                                           are 31.8 KB of LDS, the most that lets a CU hold FIVE workgroups (a CU hands LDS out in pieces
                                           that put the limit at 32 000 bytes, not 32 768: 896 rows = 32 744 bytes ran as four); the kernel's
                                           registers are budgeted to match (BATCH_WGS).  1024 rows, four a CU: k_bwt_batch 36.0 ms per 10^9
                                           bytes of wiki; 832, five: 33.3; 768 / 704, five: 35.4 / 36.0 (profiles/r06_t_ab*.txt) */
#endif
#define SMALL_BLOCK BATCH_ROWS           /* blocks of at most one batch of k_bwt_batch are sorted whole in LDS, without a partition */
#ifndef COUNT_GROUP
#define COUNT_GROUP 128u                /* groups this short are ordered by counting */
#endif
#ifndef CHUNK_WIN
#define CHUNK_WIN 128u                  /* rows per window a wave claims at a time inside a batch */
#endif
#ifndef WAVE_GROUP
#define WAVE_GROUP 1024u                /* batches holding a longer group are sorted by the whole workgroup */
#endif
#define MAX_SYMS 32u                    /* symbols per key, capped (halo of the text tile) */
#ifndef SPLIT_MIN
#define SPLIT_MIN 256u                  /* a group longer than this is first cut into sub-groups on its next key byte */
#endif
#ifndef BIG_RUN
#define BIG_RUN 63u                     /* a run of equal keys longer than this is refined in LDS (k_bwt_batch: its wave radix-sorts it on the next
                                           sy symbols of the text); shorter runs fit one 64-lane strip and go to the text rounds (k_bwt_deep) */
#endif
#ifndef BIG_ROUNDS
#define BIG_ROUNDS 16u                  /* such refinements a chunk gets at most; what is still tied in long runs then is left to the rank rounds */
#endif
#define DEEP_ROUNDS LBZ_DEEP_ROUNDS      /* launches of k_bwt_deep (lbz_kernels.h); what they leave tied goes to the rank rounds (k_bwt_fix*) */
#define DEEP_HANDOVER LBZ_DEEP_HANDOVER /* the launch of k_bwt_deep at which a block with too many rows still tied is handed to the rank rounds */
#define DEEP_BUILD LBZ_DEEP_BUILD       /* the rows still tied at the end of this launch of k_bwt_deep get rank entries: later launches may step by ranks */
#define DEEP_STACK 48u
#define DEEP_LEVELS 4u                  /* symbols a long run is split on in one launch */
#define DEEP_CHUNK 256u                 /* list entries a wave claims at a time in k_bwt_deep */
#define DEEP_STEP 13u                   /* symbols a text step decides: 2 x 52 bits of the 16 bytes it loads */
#define TIE_FLAG 0x80000000u
/* a suffix-array / tie-list entry: rotation index (n < 2^20) | dense code of the byte before it << 20 | TIE_FLAG.
   The byte rides along so that a row that becomes unique needs no look-up in the text.          */
#define SA_IDX(v) ((v) & 0x000FFFFFu)
#define SA_CODE(v) (((v) >> 20) & 255u)
#define SA_ENTRY(idx, code) ((idx) | ((u32)(code) << 20))
/* a rank entry: the rotation's rank now, its rank before the launch that wrote the entry, and that launch's tag
   (0 = k_bwt_batch / k_bwt_fix0, r + 1 = deep-tie round r).  A round reads ranks of rotations whose runs other
   workgroups of the same launch are splitting; mixing ranks from before and after a split of ONE run would order
   two rows wrongly (an old rank is only a lower bound of the new one).  With both values in one 8-byte word -- one
   store, one gather, the same 64-byte transaction as a 4-byte rank -- a reader always takes the rank as it stood when
   the launch began, whenever the word was written.                                                           */
#define ISA_ENTRY(cur, prev, tag) ((u64)(cur) | ((u64)(prev) << 20) | ((u64)(tag) << 40))
#define ISA_CUR(e) ((u32)(e) & 0x000FFFFFu)
#define ISA_TAG(e) ((u32)((e) >> 40) & 31u)
/* the text rounds (k_bwt_deep) also note, in the 19 bits above the tag, how many symbols the rotation's run shared when the
   entry was written (halved, rounded down: a lower bound) */
#define ISA_ENTRY_D(cur, prev, tag, depth) (ISA_ENTRY(cur, prev, tag) | ((u64)((depth) >> 1) << 45))
#define ISA_DEPTH(e) ((u32)((e) >> 45) << 1)
__device__ __forceinline__ u32 isa_before(u64 e, u32 tag)      /* the rank as of the start of the launch with this tag */
{
  return ISA_TAG(e) == tag ? (u32)(e >> 20) & 0x000FFFFFu : (u32)e & 0x000FFFFFu;
}
static_assert(BATCH_CAP == LBZ_WG * 4u && BATCH_ROWS <= BATCH_CAP && BATCH_ROWS % 64u == 0u, "a batch is at most one 4-rows-per-thread tile, whole strips");

#define BIG_FRAMES 14u                  /* big_group: BIG_LEVELS + 1 frames, rounded up */
struct sort_core {                      /* HBM radix passes (partition, oversized groups, doubling): what a tile's scatter needs */
  u32 wcnt[LBZ_NW][256];
  u32 dbase[256];                       /* running global offset of every digit */
  u32 toff[256];                        /* digit offsets inside the current tile */
  u32 gdelta[256];                      /* global offset minus tile offset */
  u32 wtot[8];
  u64 stage_k[SORT_TILE];               /* the tile, regrouped by digit, before it is written */
  u32 stage_v[SORT_TILE];
  u32 tile[(SORT_TILE + PART_HALO + 16u) / 4u];
};
struct sort_lds : sort_core {           /* ... and the digit histograms of the sorters that count before they scatter (k_bwt_scat does without: a
                                           fourth workgroup per CU) */
  u32 hist[8][256];
};
struct batch_lds {                      /* one batch resident in LDS */
  u64 kA[BATCH_ROWS], kB[BATCH_ROWS];
  u32 vA[BATCH_ROWS], vB[BATCH_ROWS];
  u16 gh[BATCH_ROWS], ghn[BATCH_ROWS];    /* local row of the run's first element */
  u16 gend[BATCH_ROWS];                  /* indexed by a run's first row: one past its last row */
  u8 tied[BATCH_ROWS];
  u32 wcnt[LBZ_NW][256];
  u32 dbase[256];
  u16 cstart[BATCH_CAP / 64u + 2u];     /* first row of the chunk that belongs to each claim window */
  u8 corder[BATCH_CAP / 64u + 2u];      /* chunks, longest first */
  u16 ctied[BATCH_CAP / 64u + 2u];      /* doubling: rows of each chunk that stay tied */
#ifdef BATCH_TICKS
  u32 bt[16];                           /* (diagnostic build) k_bwt_batch: ticks by phase, summed over the waves; tests/tools/quickperf.py */
#endif
};
struct bwt_lds {
  wg_scratch sc;
  u32 bc[16];
  u32 listn, seglo;                   /* k_bwt_batch: entries in the segment's list of tied rows so far; the segment's first row */
  u32 h0min, lmin;                    /* k_bwt_batch: least depth of a tie left for the rank rounds; of a run in the list */
  u32 cmin, cpad;                     /* k_bwt_batch: least depth of a closed run (wave_finish_chunk) */
  u32 fr_end[BIG_FRAMES], fr_dep[BIG_FRAMES];   /* k_bwt_batch, big_group: where each level of re-keyed rows ends, how deep its keys reach */
  u32 msd_shift, seghi;               /* 64 - the block's partition depth: rows with equal key >> msd_shift form a group; k_bwt_batch: the segment's end */
  u32 dbg[4];                         /* LDS_SORT_TICKS: whole-workgroup batch sorts (count, ticks), oversized groups (count, ticks) */
  u8 cmap[256];                       /* byte -> dense code */
  u8 inv[256];                        /* dense code -> byte */
  union {
    sort_lds X;
    batch_lds B;
  } u;
};

struct bwt_slot {
  u64 *k0, *k1;
  u32 *v0, *v1, *sufx, *grp, *pos, *sa;
  u64 *isa;
  u32 *ph;                              /* the partition's per-range digit histograms: 4 tables of PART_MAX x 256 (64 KB of the slot's tail) */
};

struct keycfg {
  u32 b;        /* bits per symbol */
  u32 sy;       /* symbols per key */
  u32 pad;      /* 64 - b*sy: keys are left aligned */
  u32 q0;       /* whole symbols in a key's top half: what the partition's 8-byte rows carry */
};
#define BATCH_DEPTH(c) ((c).q0 + 4u)    /* symbols a key of k_bwt_batch covers: the top half's, and four bytes of text behind them (row_key) */

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
  s.isa = (u64 *)p; p += (size_t)cap * 8u;
  s.ph = (u32 *)p;
  return s;
}

/* workspace of this workgroup in a round of `count` slabs (lbz_kernels.h) */
__device__ __forceinline__ bwt_slot round_slot(u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, u32 count, lbz_layout L, u32 i)
{
  return i < count ? slot_carve(ws + (u64)i * slot_bytes, L.cap_a)
                   : slot_carve(ws_spill + (u64)(i - count) * spill_bytes, L.cap_b);
}

/* the slot as a segment sees it: its list of tied rows and its scratch start at the segment's first row */
__device__ __forceinline__ bwt_slot seg_view(bwt_slot s, u32 lo)
{
  s.sufx += lo; s.grp += lo; s.pos += lo;          /* the segment's list */
  s.k0 += lo; s.k1 += lo; s.v0 += lo; s.v1 += lo;  /* scratch of the HBM sorter (runs longer than a batch) */
  return s;
}

struct sort_core;
template <bool NEXT = false, bool HALF = false>
__device__ __forceinline__ void radix_tile_scatter_hbm(sort_core *X, const unsigned long long (&key)[4], const unsigned int (&val)[4],
                                                       unsigned int okmask, unsigned int shift,
                                                       void *kout, unsigned int *vout,
                                                       unsigned int (*nh)[256] = nullptr, unsigned int nshift = 0u);

/* ======================================================================= HBM radix sorter
 * Stable LSD radix sort of m (key,value) pairs on key bits [0, nbits).  Input in (k0,v0);
 * returns 0 if the sorted result is in (k0,v0), 1 if in (k1,v1).                        */
/* (always inlined, like every device function of this library: see the note at lds_radix_sort) */
__device__ __forceinline__ u32 wg_radix_sort(u64 *k0, u32 *v0, u64 *k1, u32 *v1, u32 m, u32 nbits, bwt_lds *S)
{
  sort_lds *G = &S->u.X;
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const u32 npass = (nbits + 7u) / 8u;

  for (u32 i = tid; i < 8u * 256u; i += LBZ_WG) (&G->hist[0][0])[i] = 0;
  __syncthreads();
  for (u32 i = tid; i < m; i += LBZ_WG) {
    const u64 key = k0[i];
    for (u32 p = 0; p < npass; p++) atomicAdd(&G->hist[p][(u32)(key >> (8u * p)) & 255u], 1u);
  }
  __syncthreads();

  u32 cur = 0;
  for (u32 p = 0; p < npass; p++) {
    const u32 shift = 8u * p;
    /* digit offsets; a digit shared by every key makes the pass a no-op */
    const u32 c = tid < 256u ? G->hist[p][tid] : 0u;
    u32 tot;
    const u32 ex = wg_excl_add(c, &tot, &S->sc);
    if (tid < 256u) G->dbase[tid] = ex;
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
      u64 key[SORT_IPT];
      u32 val[SORT_IPT], okmask = 0;
      const u32 wbase = t0 + w * 64u * SORT_IPT;
#pragma unroll
      for (u32 k = 0; k < SORT_IPT; k++) {
        const u32 i = wbase + k * 64u + lane;
        key[k] = i < m ? kin[i] : 0ull;
        val[k] = i < m ? vin[i] : 0u;
        if (i < m) okmask |= 1u << k;
      }
      radix_tile_scatter_hbm<false, false>(G, key, val, okmask, shift, kout, vout);
    }
    __syncthreads();
    cur ^= 1u;
  }
  return cur;
}

/* Turn a sorted list into groups.  For list entry k (row pos_k of the suffix array):
 *   head  = starts a group: FLAGS ? the value's tie flag is clear : key differs from entry k-1
 *   rank  = row of the group's first entry
 * writes sa[row] = suffix, isa[suffix] = rank, and compacts the entries that are still
 * tied into (sufx, grp, pos) from index out_base on.  Entry k sits in row rowbase + k.  If bwt
 * is given, entries that are no longer tied get their BWT byte written.  Returns out_base plus
 * the number of still-tied entries.                                                       */
template <bool FLAGS>
__device__ u32 wg_regroup(const u64 *key, const u32 *val, u32 rowbase, u32 out_base, u32 m,
                          bwt_slot s, bwt_lds *S, const u8 *T, u32 n, u8 *bwt, u32 isa_below = 0xFFFFFFFFu,
                          u32 oldrank = 0u, u32 tag = 0u)          /* !FLAGS: the run's rank so far and the round's tag */
{
  const u32 tid = threadIdx.x;
  u32 carry_rank = 0, carry_cnt = out_base;
  for (u32 t0 = 0; t0 < m; t0 += SORT_TILE) {
    const u32 k0 = t0 + tid * SORT_IPT;
    u32 vv[SORT_IPT], row[SORT_IPT];
    u32 headmask = 0, nextmask = 0;      /* bit i: entry k0+i / entry k0+i+1 starts a group */
    if (FLAGS) {
#pragma unroll
      for (u32 i = 0; i < SORT_IPT; i++) {
        const u32 k = k0 + i;
        vv[i] = k < m ? val[k] : 0u;
        row[i] = rowbase + k;
      }
      const u32 vnext = (k0 + SORT_IPT < m) ? val[k0 + SORT_IPT] : 0u;
#pragma unroll
      for (u32 i = 0; i < SORT_IPT; i++) {
        const u32 k = k0 + i;
        if (k < m && !(vv[i] & TIE_FLAG)) headmask |= 1u << i;
        const u32 vn = (i + 1u < SORT_IPT) ? vv[(i + 1u) % SORT_IPT] : vnext;
        if (k + 1u >= m || !(vn & TIE_FLAG)) nextmask |= 1u << i;
        vv[i] &= 0x0FFFFFFFu;                                  /* index + code of the preceding byte */
      }
    } else {
      u64 kk[SORT_IPT + 2];
#pragma unroll
      for (u32 i = 0; i < SORT_IPT; i++) {
        const u32 k = k0 + i;
        kk[i + 1] = k < m ? key[k] : 0ull;
        vv[i] = k < m ? val[k] : 0u;
        row[i] = rowbase + k;
      }
      kk[0] = (k0 > 0 && k0 <= m) ? key[k0 - 1] : 0ull;
      kk[SORT_IPT + 1] = (k0 + SORT_IPT < m) ? key[k0 + SORT_IPT] : 0ull;
#pragma unroll
      for (u32 i = 0; i < SORT_IPT; i++) {
        const u32 k = k0 + i;
        if (k < m && (k == 0 || kk[i + 1] != kk[i])) headmask |= 1u << i;
        if (k + 1u >= m || kk[i + 2] != kk[i + 1]) nextmask |= 1u << i;
      }
    }
    u32 lastrank = 0, nact = 0, actmask = 0;
#pragma unroll
    for (u32 i = 0; i < SORT_IPT; i++) {
      const u32 k = k0 + i;
      if (k < m) {
        if ((headmask >> i) & 1u) lastrank = row[i] + 1u;
        /* entry k is still tied unless it and its successor both start a group */
        if (!(((headmask >> i) & 1u) && ((nextmask >> i) & 1u))) { actmask |= 1u << i; nact++; }
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
        if (!FLAGS) s.sa[row[i]] = vv[i];
        if (row[i] < isa_below)                                          /* rows from isa_below on got their rank in k_bwt_batch */
          s.isa[SA_IDX(vv[i])] = FLAGS ? ISA_ENTRY(rank1 - 1u, rank1 - 1u, 0u) : ISA_ENTRY(rank1 - 1u, oldrank, tag);
        if ((actmask >> i) & 1u) {
          s.sufx[o] = vv[i];
          s.grp[o] = rank1 - 1u;
          s.pos[o] = row[i];
          o++;
        } else if (bwt) {
          bwt[row[i]] = S->inv[SA_CODE(vv[i])];
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

/* ======================================================================= keys */
/* The sy symbols starting at rotation `start`, packed b bits each, left aligned. */
__device__ __forceinline__ u64 key_from_text(const u8 *T, u32 n, u32 start, const u8 *cmap, keycfg c)
{
  u64 key = 0;
  if (c.b == 8u && start + 8u <= n) {
    /* more than 128 byte values in use: codes are the bytes themselves (bwt_setup), the key is the text, big-endian */
    return __builtin_bswap64(reinterpret_cast<const lbz_text16 *>(T + start)->a);
  }
  if (c.sy <= 16u && start + 20u <= n) {
    /* five aligned dwords cover the 16 bytes; one wait instead of sy dependent loads */
    const u32 shb = (start & 3u) * 8u;
    const u32 *p = reinterpret_cast<const u32 *>(T + (start & ~3u));
    const u32 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4];
    const u64 lo = (u64)w0 | ((u64)w1 << 32), mid = (u64)w2 | ((u64)w3 << 32), hi = (u64)w4;
    const u64 x0 = shb ? (lo >> shb) | (mid << (64u - shb)) : lo;
    const u64 x1 = shb ? (mid >> shb) | (hi << (64u - shb)) : mid;
    for (u32 k = 0; k < c.sy; k++) {
      const u32 byte = (u32)((k < 8u ? x0 : x1) >> (8u * (k & 7u))) & 255u;
      key = (key << c.b) | cmap[byte];
    }
  } else {
    u32 j = start;
    for (u32 k = 0; k < c.sy; k++) {
      key = (key << c.b) | cmap[T[j]];
      j = (j + 1u == n) ? 0u : j + 1u;
    }
  }
  return key << c.pad;
}

/* The key k_bwt_batch orders a row by: the top half the partition carried (q0 whole symbols, and perhaps some bits of the
 * next), and behind it the four BYTES of text from symbol q0 on.  Bytes, not codes: the dense codes keep the bytes' order, so
 * inside a group -- rows whose top halves agree -- bytes order the rows exactly as the codes would, and a 4-byte load replaces a
 * table look-up per symbol.  Two rows with the same key share q0 + 4 symbols (BATCH_DEPTH). */
__device__ __forceinline__ u64 row_key(const u8 *T, u32 n, u32 hi32, u32 idx, keycfg c)
{
  u32 at = idx + c.q0;
  if (at >= n) at -= n;
  u32 raw;
  if (at + 4u <= n) raw = __builtin_bswap32(reinterpret_cast<const lbz_text4 *>(T + at)->a);
  else {
    raw = 0;
    for (u32 q = 0; q < 4u; q++) { raw = (raw << 8) | T[at]; at = at + 1u == n ? 0u : at + 1u; }
  }
  return ((u64)hi32 << 32) | raw;
}

/* One stable counting-sort step of a 4096-row tile on digit (key >> shift) & 255: rows are
 * held wave-striped (row = wbase + k*64 + lane), ranks inside a wave come from ballots, the
 * per-wave counters and the running digit offsets (dbase) live in LDS.  Its barriers order
 * LDS only: the scattered rows are not read again before the next full barrier.          */
__device__ __forceinline__ void radix_tile_scatter(u32 (*wcnt)[256], u32 *dbase, const u64 (&key)[SORT_IPT],
                                                   const u32 (&val)[SORT_IPT], u32 okmask, u32 shift,
                                                   u64 *kout, u32 *vout)
{
  const u32 tid = threadIdx.x, w = wave_id();
  u32 rnk[SORT_IPT];
  for (u32 i = tid; i < LBZ_NW * 256u; i += LBZ_WG) (&wcnt[0][0])[i] = 0;
  wg_lds_barrier();
#pragma unroll
  for (u32 k = 0; k < SORT_IPT; k++) {
    const bool ok = (okmask >> k) & 1u;
    const u32 d = (u32)(key[k] >> shift) & 255u;
    const u64 mask = match_digit(d, ok);
    const u32 below = (u32)__popcll(mask & lanes_below());
    const u32 prev = ok ? wcnt[w][d] : 0u;
    wave_sync();
    if (ok && below == 0u) wcnt[w][d] = prev + (u32)__popcll(mask);
    wave_sync();
    rnk[k] = prev + below;
  }
  wg_lds_barrier();
  if (tid < 256u) {
    u32 run = dbase[tid];
#pragma unroll
    for (u32 w2 = 0; w2 < LBZ_NW; w2++) {
      const u32 t = wcnt[w2][tid];
      wcnt[w2][tid] = run;
      run += t;
    }
    dbase[tid] = run;
  }
  wg_lds_barrier();
#pragma unroll
  for (u32 k = 0; k < SORT_IPT; k++) {
    if ((okmask >> k) & 1u) {
      const u32 d = (u32)(key[k] >> shift) & 255u;
      const u32 dst = wcnt[w][d] + rnk[k];
      kout[dst] = key[k];
      vout[dst] = val[k];
    }
  }
  wg_lds_barrier();
}

/* The same step for an HBM destination: ranks as above, then the tile is regrouped by digit
 * in LDS and written out by consecutive threads, so that each wave store covers a few runs of
 * consecutive addresses instead of 64 scattered rows.  NEXT is a template argument on purpose: as a run-time test inside
 * the write-out loop it cost the array passes 35 % (3.4 ms against 2.45 per pass of 371 blocks, r04 traces).      */
/* HALF (round 6): the partition's rows are 8 bytes -- the key's TOP HALF and the value.  Its passes read digits of the top 32
 * bits only; the low half of a key is first looked at by k_bwt_batch, which rebuilds it from the block's text (in its XCD's L2
 * by then) instead of carrying it four times through HBM: 56 bytes of payload per rotation instead of 84. */
template <bool NEXT, bool HALF>
__device__ __forceinline__ void radix_tile_scatter_hbm(sort_core *X, const u64 (&key)[SORT_IPT],
                                                       const u32 (&val)[SORT_IPT], u32 okmask, u32 shift,
                                                       void *kout_, u32 *vout, u32 (*nh)[256], u32 nshift)
{
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  u32 rnk[SORT_IPT];
  for (u32 i = tid; i < LBZ_NW * 256u; i += LBZ_WG) (&X->wcnt[0][0])[i] = 0;
  wg_lds_barrier();
#pragma unroll
  for (u32 k = 0; k < SORT_IPT; k++) {
    const bool ok = (okmask >> k) & 1u;
    const u32 d = (u32)(key[k] >> shift) & 255u;
    const u64 mask = match_digit(d, ok);
    const u32 below = (u32)__popcll(mask & lanes_below());
    const u32 prev = ok ? X->wcnt[w][d] : 0u;
    wave_sync();
    if (ok && below == 0u) X->wcnt[w][d] = prev + (u32)__popcll(mask);
    wave_sync();
    rnk[k] = prev + below;
  }
  wg_lds_barrier();
  u32 tot = 0, inc = 0;
  if (tid < 256u) {
#pragma unroll
    for (u32 w2 = 0; w2 < LBZ_NW; w2++) {
      const u32 t = X->wcnt[w2][tid];
      X->wcnt[w2][tid] = tot;
      tot += t;
    }
    inc = wave_incl_add(tot);
    if (lane == 63u) X->wtot[w] = inc;
  }
  wg_lds_barrier();
  if (tid < 256u) {
    u32 base = 0;
    for (u32 w2 = 0; w2 < w; w2++) base += X->wtot[w2];
    const u32 toff = base + inc - tot;
    const u32 g = X->dbase[tid];
    X->toff[tid] = toff;
    X->gdelta[tid] = g - toff;
    X->dbase[tid] = g + tot;
  }
  wg_lds_barrier();
#pragma unroll
  for (u32 k = 0; k < SORT_IPT; k++) {
    if ((okmask >> k) & 1u) {
      const u32 d = (u32)(key[k] >> shift) & 255u;
      const u32 lpos = X->toff[d] + X->wcnt[w][d] + rnk[k];
      if (HALF) reinterpret_cast<u32 *>(X->stage_k)[lpos] = (u32)(key[k] >> 32);
      else X->stage_k[lpos] = key[k];
      X->stage_v[lpos] = val[k];
    }
  }
  wg_lds_barrier();
  const u32 rows = X->wtot[0] + X->wtot[1] + X->wtot[2] + X->wtot[3];
#pragma unroll
  for (u32 k = 0; k < SORT_IPT; k++) {
    const u32 i = tid + k * LBZ_WG;
    if (i < rows) {
      if (HALF) {
        const u32 kk = reinterpret_cast<const u32 *>(X->stage_k)[i];
        const u32 dst = X->gdelta[(kk >> (shift - 32u)) & 255u] + i;
        stg_u32(reinterpret_cast<u32 *>(kout_) + dst, kk);
        stg_u32(vout + dst, X->stage_v[i]);
        if (NEXT) atomicAdd(&nh[dst >> nshift][(kk >> (shift - 24u)) & 255u], 1u);
      } else {
        const u64 kk = X->stage_k[i];
        const u32 dst = X->gdelta[(u32)(kk >> shift) & 255u] + i;
        stg_u64(reinterpret_cast<u64 *>(kout_) + dst, kk);
        stg_u32(vout + dst, X->stage_v[i]);
        if (NEXT) atomicAdd(&nh[dst >> nshift][(u32)(kk >> (shift + 8u)) & 255u], 1u);      /* the NEXT pass's digit, by the range of its input the row lands in */
      }
    }
  }
  wg_lds_barrier();
}

/* A pass over the block text through the LDS tile, keys built on the fly.  The tile holds
 * dense symbol CODES (one table lookup per text byte); a thread owns 4 consecutive rotations,
 * reads their 4+sy-1 codes as five dwords and slides a window over them.
 * SCATTER = false: histogram of the partition digit at key bit 32 of every rotation (hist[0]);
 * SCATTER = true : first partition pass (digit at `shift`) straight from the text, counting the digits above it on
 *                  the way if `count_above` (hist[j]: the digit at bit 32 + 8 j).  Values carry the CODE of the
 *                  preceding byte; the emitters map it back.                                 */
template <bool SCATTER, bool NEXT = false>
__device__ __forceinline__ void msd_text_pass(const u8 *T, u32 n, keycfg c, u32 *kout, u32 *vout, bwt_lds *S, u32 shift, u32 r0, u32 r1,
                              u32 (*nh)[256] = nullptr, u32 nshift = 0u)      /* rotations [r0, r1), r0 a multiple of the tile */
{
  sort_lds *P = &S->u.X;
  const u32 tid = threadIdx.x;
  u32 *tile32 = P->tile;        /* code of position t0 + j at byte 4 + j */
  const u64 keep = (c.b * c.sy >= 64u) ? ~0ull : ((1ull << (c.b * c.sy)) - 1ull);
  /* a tile's text: four bytes per thread, the halo behind the tile (a few threads) and the byte before it (one thread),
     requested one tile ahead and left in flight across the ranking of the tile before */
  u32 raw = 0, halo = 0, before = 0;
  auto fetch = [&](u32 t0) {
    const u32 q0 = t0 + 4u * tid;
    if (q0 + 4u <= n) raw = ldg_u32(reinterpret_cast<const u32 *>(T + q0));
    else raw = ldg_u8(T + q0 % n) | (ldg_u8(T + (q0 + 1u) % n) << 8) | (ldg_u8(T + (q0 + 2u) % n) << 16) | (ldg_u8(T + (q0 + 3u) % n) << 24);
    if (tid < PART_HALO / 4u) {
      const u32 h0 = t0 + SORT_TILE + 4u * tid;
      if (h0 + 4u <= n) halo = ldg_u32(reinterpret_cast<const u32 *>(T + h0));
      else halo = ldg_u8(T + h0 % n) | (ldg_u8(T + (h0 + 1u) % n) << 8) | (ldg_u8(T + (h0 + 2u) % n) << 16) | (ldg_u8(T + (h0 + 3u) % n) << 24);
    }
    if (tid == PART_HALO / 4u) before = ldg_u8(T + (t0 + n - 1u) % n);
  };
  auto codes = [&](u32 x) {
    return (u32)S->cmap[x & 255u] | ((u32)S->cmap[(x >> 8) & 255u] << 8) | ((u32)S->cmap[(x >> 16) & 255u] << 16) | ((u32)S->cmap[x >> 24] << 24);
  };
  if (r0 < r1) fetch(r0);
  for (u32 t0 = r0; t0 < r1; t0 += SORT_TILE) {
    const u32 q0 = t0 + 4u * tid;
    tile32[1u + tid] = codes(raw);
    if (tid < PART_HALO / 4u) tile32[1u + LBZ_WG + tid] = codes(halo);
    if (tid == PART_HALO / 4u) tile32[0] = (u32)S->cmap[before] << 24;
    fetch(t0 + SORT_TILE);
    __syncthreads();
    u64 key[SORT_IPT];
    u32 val[SORT_IPT], okmask = 0;
    if (c.sy <= 16u) {
      const u32 w0 = tile32[1u + tid], w1 = tile32[2u + tid], w2 = tile32[3u + tid], w3 = tile32[4u + tid], w4 = tile32[5u + tid];
      const u64 x0 = (u64)w0 | ((u64)w1 << 32), x1 = (u64)w2 | ((u64)w3 << 32), x2 = (u64)w4;
      u64 win = 0;
      for (u32 q = 0; q < c.sy; q++)
        win = (win << c.b) | (u32)(((q < 8u ? x0 : x1) >> (8u * (q & 7u))) & 255ull);
      u32 prev = tile32[tid] >> 24;
#pragma unroll
      for (u32 k = 0; k < SORT_IPT; k++) {
        if (k) {
          const u32 q = k + c.sy - 1u;                   /* <= 18 */
          const u32 code = (u32)(((q < 8u ? x0 : (q < 16u ? x1 : x2)) >> (8u * (q & 7u))) & 255ull);
          win = ((win << c.b) | code) & keep;
        }
        key[k] = win << c.pad;
        val[k] = (prev << 24) | (q0 + k);
        prev = (w0 >> (8u * k)) & 255u;
        if (q0 + k < r1) okmask |= 1u << k;
      }
    } else {
#pragma unroll
      for (u32 k = 0; k < SORT_IPT; k++) {
        const u32 j = 4u * tid + k;
        u64 kk = 0;
        for (u32 q = 0; q < c.sy; q++) kk = (kk << c.b) | reinterpret_cast<const u8 *>(P->tile)[4u + j + q];
        key[k] = kk << c.pad;
        val[k] = ((u32)reinterpret_cast<const u8 *>(P->tile)[3u + j] << 24) | (q0 + k);
        if (q0 + k < r1) okmask |= 1u << k;
      }
    }
    if (SCATTER) {
      radix_tile_scatter_hbm<NEXT, true>(P, key, val, okmask, shift, kout, vout, nh, nshift);
    } else {
      /* the first pass's digit of every rotation of the range, for either depth of partition: hist[0] the key byte at bit 32
         (32-bit partition), hist[2] the one at bit 48 (16-bit) */
#pragma unroll
      for (u32 k = 0; k < SORT_IPT; k++)
        if ((okmask >> k) & 1u) {
          atomicAdd(&P->hist[0][(u32)(key[k] >> 32u) & 255u], 1u);
          atomicAdd(&P->hist[2][(u32)(key[k] >> 48u) & 255u], 1u);
        }
      __syncthreads();
    }
  }
}

/* A partition pass from (kin,vin) to (kout,vout) on the digit at `shift`.  The next tile's rows are requested before the
 * current tile is ranked, and stay in flight across the ranking: the loads are unconditional (rows past the end read the last
 * row again; okmask drops them) and say "global" (lbz_asm.h), so no wait for an LDS read waits for them.  The compiler's own
 * form of this loop -- a guarded flat load per row, each followed by its wait -- ran a pass of 371 blocks in 3.3 ms. */
template <bool NEXT>
__device__ __forceinline__ void msd_array_pass(const u32 *kin, const u32 *vin, u32 shift, u32 *kout, u32 *vout, bwt_lds *S, u32 r0, u32 r1,
                                               u32 (*nh)[256], u32 nshift)      /* rows [r0, r1) */
{
  if (r0 >= r1) return;                                      /* (workgroup-uniform) */
  sort_lds *P = &S->u.X;
  const u32 lane = lane_id(), w = wave_id();
  const u32 last = r1 - 1u;
  u32 nkey[SORT_IPT];
  u32 nval[SORT_IPT];
#pragma unroll
  for (u32 k = 0; k < SORT_IPT; k++) {
    const u32 i0 = r0 + w * 64u * SORT_IPT + k * 64u + lane, i = i0 < last ? i0 : last;
    nkey[k] = ldg_u32(kin + i);
    nval[k] = ldg_u32(vin + i);
  }
  for (u32 t0 = r0; t0 < r1; t0 += SORT_TILE) {
    u64 key[SORT_IPT];
    u32 val[SORT_IPT], okmask = 0;
    const u32 wbase = t0 + w * 64u * SORT_IPT;
#pragma unroll
    for (u32 k = 0; k < SORT_IPT; k++) {
      key[k] = (u64)nkey[k] << 32;
      val[k] = nval[k];
      if (wbase + k * 64u + lane < r1) okmask |= 1u << k;
    }
#pragma unroll
    for (u32 k = 0; k < SORT_IPT; k++) {
      const u32 i0 = wbase + SORT_TILE + k * 64u + lane, i = i0 < last ? i0 : last;
      nkey[k] = ldg_u32(kin + i);
      nval[k] = ldg_u32(vin + i);
    }
    radix_tile_scatter_hbm<NEXT, true>(P, key, val, okmask, shift, kout, vout, nh, nshift);
  }
}

/* exclusive offsets of one digit histogram into dbase */
__device__ __forceinline__ void load_digit_offsets(const u32 *hist, u32 *dbase, bwt_lds *S)
{
  const u32 tid = threadIdx.x;
  u32 tot;
  const u32 ex = wg_excl_add(tid < 256u ? hist[tid] : 0u, &tot, &S->sc);
  if (tid < 256u) dbase[tid] = ex;
  __syncthreads();
}

/* ======================================================================= LDS batch */
/* Radix sort of cnt <= BATCH_CAP pairs held in (kA,vA); returns 0/1 = result in A/B. */
/* NO DEVICE FUNCTION CALLS.  Left to itself the compiler kept this function and wg_radix_sort out of line in k_bwt_batch (two
 * s_swappc call sites each, 67 scalar registers of the kernel spilled to lanes of v127 around them).  In round 6 a build of that
 * shape sorted wiki blocks wrongly on the device -- rows unwritten behind the first oversized group, differently from run to run,
 * with one workgroup per block as with 32 -- while the emulator, every sanitizer run and three builds that differed by one
 * never-taken bounds check were right; with the two functions inlined every one of those builds is right (tests/tools/dbg_bwt.py,
 * profiles/r06_g_*).  The kernels of this library therefore contain no calls: `make calls` (csrc/Makefile) fails the build if
 * a code object has an s_swappc. */
__device__ __forceinline__ u32 lds_radix_sort(batch_lds *B, u32 cnt, bwt_lds *S)
{
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  u32 *dbase = B->dbase;
  /* which key bytes differ between any two keys of the batch? */
  u64 vo = 0, va = ~0ull;
  for (u32 i = tid; i < cnt; i += LBZ_WG) { const u64 k = B->kA[i]; vo |= k; va &= k; }
  u64 ro, ra;
  wg_or_and64(vo, va, &ro, &ra, &S->sc);
  const u64 varying = ro ^ ra;

  u32 cur = 0;
  for (u32 p = 0; p < 8u; p++) {
    const u32 shift = 8u * p;
    if (((varying >> shift) & 255ull) == 0ull) continue;
    const u64 *kin = cur ? B->kB : B->kA;
    const u32 *vin = cur ? B->vB : B->vA;
    u64 key[SORT_IPT];
    u32 val[SORT_IPT], okmask = 0;
    const u32 wbase = w * 64u * SORT_IPT;
#pragma unroll
    for (u32 k = 0; k < SORT_IPT; k++) {
      const u32 i = wbase + k * 64u + lane;
      key[k] = i < cnt ? kin[i] : 0ull;
      val[k] = i < cnt ? vin[i] : 0u;
      if (i < cnt) okmask |= 1u << k;
    }
    if (tid < 256u) dbase[tid] = 0;
    __syncthreads();
    /* digit totals -> exclusive offsets: count first, then the ordinary tile scatter */
#pragma unroll
    for (u32 k = 0; k < SORT_IPT; k++)
      if ((okmask >> k) & 1u) atomicAdd(&dbase[(u32)(key[k] >> shift) & 255u], 1u);
    __syncthreads();
    load_digit_offsets(dbase, dbase, S);
    radix_tile_scatter(B->wcnt, dbase, key, val, okmask, shift, cur ? B->kA : B->kB, cur ? B->vA : B->vB);
    cur ^= 1u;
  }
  return cur;
}

/* Wave-private LSD radix sort of rows [cs, ce) of the batch (data in A, result in A): strips
 * of 64 rows, the wave's own digit counters, no workgroup barrier.  Key bytes that do not
 * vary inside the range are skipped, so a group that shares its top 24 bits costs <= 5 passes. */
template <bool TOP_ONLY = false>
__device__ u32 wave_radix_range(batch_lds *B, u32 cs, u32 ce)
{
  const u32 lane = lane_id(), w = wave_id();
  u64 vo = 0, va = ~0ull;
  for (u32 j = cs + lane; j < ce; j += 64u) { const u64 k = B->kA[j]; vo |= k; va &= k; }
  wave_or_and64(&vo, &va);
  const u64 varying = vo ^ va;
  if (TOP_ONLY && varying == 0ull) return 64u;
  const u32 ptop = TOP_ONLY ? (63u - (u32)__clzll((long long)varying)) / 8u : 0u;   /* most significant varying byte */
  u32 cur = 0;
  for (u32 p = 0; p < 8u; p++) {
    const u32 shift = 8u * p;
    if (((varying >> shift) & 255ull) == 0ull) continue;
    if (TOP_ONLY && p != ptop) continue;
    const u64 *kin = cur ? B->kB : B->kA;
    const u32 *vin = cur ? B->vB : B->vA;
    u64 *kout = cur ? B->kA : B->kB;
    u32 *vout = cur ? B->vA : B->vB;
    u32 *cntw = B->wcnt[w];
#pragma unroll
    for (u32 i = 0; i < 4u; i++) cntw[lane + 64u * i] = 0;
    wave_sync();
    for (u32 j0 = cs; j0 < ce; j0 += 128u) {            /* ranks of equal digits, in row order; two strips a trip: their keys are read
                                                           and their digits matched together, the counters take them one after the other */
      u32 d2[2], below2[2], tot2[2];
      bool ok2[2];
#pragma unroll
      for (u32 q = 0; q < 2u; q++) {
        const u32 j = j0 + 64u * q + lane;
        ok2[q] = j < ce;
        d2[q] = (u32)(kin[ok2[q] ? j : cs] >> shift) & 255u;
      }
#pragma unroll
      for (u32 q = 0; q < 2u; q++) {
        const u64 mask = match_digit(d2[q], ok2[q]);
        below2[q] = (u32)__popcll(mask & lanes_below());
        tot2[q] = (u32)__popcll(mask);
      }
#pragma unroll
      for (u32 q = 0; q < 2u; q++) {
        const u32 j = j0 + 64u * q + lane;
        const u32 prev = cntw[d2[q]];
        wave_sync();
        if (ok2[q] && below2[q] == 0u) cntw[d2[q]] = prev + tot2[q];
        wave_sync();
        if (ok2[q]) B->ghn[j] = (u16)(prev + below2[q]);
      }
    }
    {                                                   /* exclusive scan of the 256 counters */
      const u32 c0 = cntw[4u * lane], c1 = cntw[4u * lane + 1u], c2 = cntw[4u * lane + 2u], c3 = cntw[4u * lane + 3u];
      const u32 sum = c0 + c1 + c2 + c3;
      const u32 ex = wave_incl_add(sum) - sum;
      wave_sync();
      cntw[4u * lane] = ex; cntw[4u * lane + 1u] = ex + c0;
      cntw[4u * lane + 2u] = ex + c0 + c1; cntw[4u * lane + 3u] = ex + c0 + c1 + c2;
      wave_sync();
    }
    for (u32 j0 = cs; j0 < ce; j0 += 128u) {
      u64 k2[2];
      u32 v2[2], r2[2], c2[2];
#pragma unroll
      for (u32 q = 0; q < 2u; q++) {
        const u32 j = j0 + 64u * q + lane, jc = j < ce ? j : cs;
        k2[q] = kin[jc]; v2[q] = vin[jc]; r2[q] = B->ghn[jc];
      }
#pragma unroll
      for (u32 q = 0; q < 2u; q++) c2[q] = cntw[(u32)(k2[q] >> shift) & 255u];
#pragma unroll
      for (u32 q = 0; q < 2u; q++) {
        const u32 j = j0 + 64u * q + lane;
        if (j < ce) {
          const u32 dst = cs + c2[q] + r2[q];
          kout[dst] = k2[q];
          vout[dst] = v2[q];
        }
      }
    }
    wave_sync();
    cur ^= 1u;
  }
  if (cur) {
    for (u32 j = cs + lane; j < ce; j += 64u) { B->kA[j] = B->kB[j]; B->vA[j] = B->vB[j]; }
    wave_sync();
  }
  return 8u * ptop;
}

#ifdef BATCH_TICKS
#define BT_MARK(v) const u64 v = wall_clock64()
#define BT_ADD(S_, i, a, b) do { if (lane_id() == 0u) atomicAdd(&(S_)->u.B.bt[i], (u32)((b) - (a))); } while (0)
#else
#define BT_MARK(v)
#define BT_ADD(S_, i, a, b)
#endif
/* Order the rows of chunk [cs, ce) (whole groups of equal top MSD_BITS, data in A) by their
 * full keys.  Rows of short groups are placed by counting the smaller keys of their group;
 * each long group is radix-sorted on its own.  Wave-private: no workgroup barrier.          */
template <bool PREFIX_EQUAL>
__device__ void wave_sort_chunk(batch_lds *B, u32 cs, u32 ce)
{
  const u32 lane = lane_id();
#ifdef BATCH_TICKS
  bwt_lds *St = reinterpret_cast<bwt_lds *>(reinterpret_cast<char *>(B) - offsetof(bwt_lds, u));
#endif
  BT_MARK(ta0);
  if (PREFIX_EQUAL) {
    /* comparison keys made unique by the row number: (key << 12) | row.  Rows of one group agree
       in the bits that are shifted out and in the 12 bits below them, so y < x is the sign of y - x */
    for (u32 j = cs + lane; j < ce; j += 64u) B->kB[j] = (B->kA[j] << 12) | (u64)j;
    wave_sync();
  }
  if (PREFIX_EQUAL) {
    /* Counted rows go straight to their place: the value into vB, the key back into kA -- rebuilt
       from the unique key (its low 52 bits) and the 12 bits every row of the group shares, so it
       does not matter that kA is being overwritten while other rows are still counting.      */
    for (u32 j0 = cs; j0 < ce; j0 += 64u) {
      const u32 j = j0 + lane;
      if (j < ce) {
        const u32 gs = B->gh[j], ge = B->gend[gs];
        const u32 v = B->vA[j];
        u32 dst = j;
        if (ge - gs <= COUNT_GROUP && ge - gs > 1u) {
          const u64 x = B->kB[j];
          const u64 top = B->kA[gs] & 0xFFF0000000000000ull;
          dst = gs;
          u32 p = gs;
#ifndef COUNT_NARROW
          for (; p + 8u <= ge; p += 8u) {                /* eight keys requested before the first is counted: the loop is LDS round trips */
            const u64 y0 = B->kB[p], y1 = B->kB[p + 1u], y2 = B->kB[p + 2u], y3 = B->kB[p + 3u];
            const u64 y4 = B->kB[p + 4u], y5 = B->kB[p + 5u], y6 = B->kB[p + 6u], y7 = B->kB[p + 7u];
            dst += (u32)((y0 - x) >> 63) + (u32)((y1 - x) >> 63) + (u32)((y2 - x) >> 63) + (u32)((y3 - x) >> 63);
            dst += (u32)((y4 - x) >> 63) + (u32)((y5 - x) >> 63) + (u32)((y6 - x) >> 63) + (u32)((y7 - x) >> 63);
          }
#endif
          for (; p + 2u <= ge; p += 2u) {
            const u64 y0 = B->kB[p], y1 = B->kB[p + 1u];
            dst += (u32)((y0 - x) >> 63) + (u32)((y1 - x) >> 63);
          }
          if (p < ge) dst += (u32)((B->kB[p] - x) >> 63);
          B->kA[dst] = top | (x >> 12);
        }
        B->vB[dst] = v;
      }
    }
    wave_sync();
    for (u32 j = cs + lane; j < ce; j += 64u) B->vA[j] = B->vB[j];
    wave_sync();
  } else {
    for (u32 j0 = cs; j0 < ce; j0 += 64u) {
      const u32 j = j0 + lane;
      if (j < ce) {
        const u32 gs = B->gh[j], ge = B->gend[gs];
        u32 dst = j;
        if (ge - gs <= COUNT_GROUP && ge - gs > 1u) {
          dst = gs;
          const u64 x = B->kA[j];
          for (u32 q = gs; q < ge; q++) {
            const u64 y = B->kA[q];
            dst += (y < x) || (y == x && q < j);
          }
        }
        B->ghn[j] = (u16)dst;
      }
    }
    wave_sync();
    for (u32 j = cs + lane; j < ce; j += 64u) {
      const u32 dst = B->ghn[j];
      B->kB[dst] = B->kA[j];
      B->vB[dst] = B->vA[j];
    }
    wave_sync();
    for (u32 j = cs + lane; j < ce; j += 64u) { B->kA[j] = B->kB[j]; B->vA[j] = B->vB[j]; }
    wave_sync();
  }
#ifdef SORT_TICKS
  const u64 tl0 = wall_clock64();
#endif
  BT_MARK(ta1);
#ifdef BATCH_TICKS
  if (PREFIX_EQUAL) {
    u32 n1 = 0, nc = 0, nl = 0;
    for (u32 j0 = cs; j0 < ce; j0 += 64u) {
      const u32 j = j0 + lane;
      const u32 g = j < ce ? (u32)B->gend[B->gh[j]] - (u32)B->gh[j] : 0u;
      n1 += (u32)__popcll(__ballot(g == 1u)); nc += (u32)__popcll(__ballot(g > 1u && g <= COUNT_GROUP)); nl += (u32)__popcll(__ballot(g > COUNT_GROUP));
    }
    if (lane == 0u) { atomicAdd(&B->bt[11], n1); atomicAdd(&B->bt[12], nc); atomicAdd(&B->bt[13], nl); }
  }
#endif
  for (u32 j0 = cs; j0 < ce; j0 += 64u) {               /* long groups, one after the other */
    const u32 j = j0 + lane;
    bool longhead = false;
    if (j < ce && B->gh[j] == j) longhead = ((u32)B->gend[j] - j) > COUNT_GROUP;
    u64 heads = __ballot(longhead);
    while (heads) {
      const u32 gs = j0 + (u32)__ffsll((long long)heads) - 1u;
      heads &= heads - 1ull;
      wave_radix_range<false>(B, gs, B->gend[gs]);
    }
  }
#ifdef SORT_TICKS
  if (PREFIX_EQUAL && lane == 0u) atomicAdd(&reinterpret_cast<bwt_lds *>(reinterpret_cast<char *>(B) - offsetof(bwt_lds, u))->bc[15], (u32)(wall_clock64() - tl0));
#endif
  BT_MARK(ta2);
  if (PREFIX_EQUAL) { BT_ADD(St, 0, ta0, ta1); BT_ADD(St, 1, ta1, ta2); }
}

/* Runs of equal 64-bit keys inside the sorted chunk [cs, ce): gh, gend, tied, wave-private.
 * WITHIN: the chunk already has runs (gh) that have just been re-sorted on new keys; new runs
 * never cross an old run's first row.  Returns the number of tied rows (wave-uniform).      */
template <bool WITHIN>
__device__ u32 wave_runs(batch_lds *B, u32 cs, u32 ce, u32 sh = 0u)
{
  const u32 lane = lane_id();
  u32 carry = cs, ntied = 0;
  for (u32 j0 = cs; j0 < ce; j0 += 128u) {               /* two strips a trip, their reads requested together (wave_finish_chunk's output loop) */
    u64 kc[2], kp[2], kn[2];
    u32 g0[2], g1[2];
#pragma unroll
    for (u32 q = 0; q < 2u; q++) {
      const u32 j = j0 + 64u * q + lane, jc = j < ce ? j : cs;
      const u32 jp = jc > cs ? jc - 1u : cs, jn = jc + 1u < ce ? jc + 1u : jc;
      kc[q] = B->kA[jc] >> sh; kp[q] = B->kA[jp] >> sh; kn[q] = B->kA[jn] >> sh;
      if (WITHIN) { g0[q] = B->gh[jc]; g1[q] = B->gh[jn]; }
    }
    wave_sync();                                         /* old gh read before it is rewritten */
#pragma unroll
    for (u32 q = 0; q < 2u; q++) {
      const u32 j = j0 + 64u * q + lane;
      const bool ok = j < ce;
      bool hd = ok && (j == cs || kp[q] != kc[q]);
      bool hn = ok && (j + 1u >= ce || kn[q] != kc[q]);
      if (WITHIN && ok) {
        hd = hd || g0[q] == j;
        hn = hn || (j + 1u < ce && g1[q] == j + 1u);
      }
      u32 h = wave_incl_max(hd ? j : 0u);
      if (h < carry) h = carry;                          /* run opened in an earlier strip */
      if (ok) {
        B->gh[j] = (u16)h;
        B->tied[j] = (hd && hn) ? 0 : 1;
        if (hn) B->gend[h] = (u16)(j + 1u);
      }
      ntied += (u32)__popcll(__ballot(ok && !(hd && hn)));
      if (j0 + 64u * q < ce) carry = (u32)__builtin_amdgcn_readlane((int)h, 63);
    }
  }
  wave_sync();
  return ntied;
}

/* Everything a wave does for its chunk after the groups are known: order the rows (unless they already are) and emit
 * them.  Rows that stay tied (runs of equal keys: on text two thirds of a block) are appended, run by run, to the
 * segment's list for the text rounds (k_bwt_deep) together with `depth`, the symbols a key covers.  No workgroup
 * barrier inside: the waves of a batch run their chunks independently.                                            */
__device__ void wave_finish_chunk(batch_lds *B, u32 cs, u32 ce, bool need_sort, u32 depth,
                                  u8 *bwt, bwt_slot s, u32 lo, lbz_block_meta *meta, bwt_lds *S)
{
  const u32 lane = lane_id();
  u32 ntied;
  const u64 tw0 = wall_clock64();
  if (need_sort) {
    wave_sort_chunk<true>(B, cs, ce);
#ifdef SORT_TICKS
    const u64 tr0 = wall_clock64();
#endif
    BT_MARK(tr0b);
    ntied = wave_runs<false>(B, cs, ce);
    BT_MARK(tr1b);
    BT_ADD(S, 2, tr0b, tr1b);
#ifdef SORT_TICKS
    if (lane == 0u) atomicAdd(&S->bc[2], (u32)(wall_clock64() - tr0));
#endif
  } else {
    ntied = 0;
    for (u32 j0 = cs; j0 < ce; j0 += 64u) {
      const u32 j = j0 + lane;
      ntied += (u32)__popcll(__ballot(j < ce && B->tied[j]));
    }
  }
  const u64 tw1 = wall_clock64();
  /* CLOSED RUNS.  What the stage owes is the byte in front of every row, in row order, and the row of rotation 0 -- not the
     order of the rows.  A run of tied rows that all have the SAME byte in front of them (and none of which is rotation 0)
     writes that byte into each of its rows whatever their order: it is finished as it stands, at any depth, and so is every
     run it could still split into.  On text two thirds of the rows tied on their first key are of that kind (a passage that
     occurs twice is tied row for row with its copy, and only where the two begin do the bytes in front differ); in a tree of
     sources nine tenths.  Such a run is not listed; its rows stay tied in the suffix array, where the rank rounds -- the one
     consumer of that array -- find them should the block need them (deep_closed).  An OPEN run is marked in its first row's
     `tied` byte (bit 1; every writer stores the same value). */
  u32 nclosed = 0;
  if (ntied) {
    /* (two strips a trip, unconditional reads, as in the output loop below: `tied` is only ever tested for "not zero" here, so
       a mark another strip sets meanwhile changes nothing) */
    for (u32 j0 = cs; j0 < ce; j0 += 128u) {
      u32 jj[2], v[2], tj[2], hd[2], vh[2];
#pragma unroll
      for (u32 q = 0; q < 2u; q++) {
        const u32 j = j0 + 64u * q + lane;
        jj[q] = j < ce ? j : cs;
        const u32 t = B->tied[jj[q]];
        tj[q] = j < ce ? t : 0u; v[q] = B->vA[jj[q]]; hd[q] = B->gh[jj[q]];
      }
#pragma unroll
      for (u32 q = 0; q < 2u; q++) vh[q] = B->vA[tj[q] ? hd[q] : jj[q]];
#pragma unroll
      for (u32 q = 0; q < 2u; q++) {
#ifdef DBG_NOCLOSE_BATCH
        if (tj[q]) B->tied[hd[q]] = 3;
#else
        if (tj[q] && ((v[q] >> 24) != (vh[q] >> 24) || (v[q] & 0x00FFFFFFu) == 0u)) B->tied[hd[q]] = 3;
#endif
      }
    }
    wave_sync();
    u32 nopen = 0;
    for (u32 j0 = cs; j0 < ce; j0 += 128u) {
      u32 tj[2], hd[2];
#pragma unroll
      for (u32 q = 0; q < 2u; q++) {
        const u32 j = j0 + 64u * q + lane, jc = j < ce ? j : cs;
        const u32 t = B->tied[jc];
        tj[q] = j < ce ? t : 0u; hd[q] = B->gh[jc];
      }
#pragma unroll
      for (u32 q = 0; q < 2u; q++) {
        const u32 th = B->tied[tj[q] ? hd[q] : cs];
        nopen += (u32)__popcll(__ballot(tj[q] && (th & 2u)));
      }
    }
    nclosed = ntied - nopen;
    ntied = nopen;
  }
  BT_MARK(tc1);
  BT_ADD(S, 3, tw1, tc1);
  const bwt_slot ls = seg_view(s, S->seglo);
  /* the chunk's tied runs take ONE stretch of the list: the rows of a run must lie side by side there */
  u32 lbase = ntied ? wave_reserve(&S->listn, ntied) : 0u;
  u32 nlong = 0;
  /* Two strips a trip, their LDS reads requested together: a strip is two dependent round trips (the row; then its run's first
     row and the byte's code) and nothing else of weight, so one strip at a time the wave mostly waits (20 of a wiki block's 62
     wave-ms were this loop).  The reads are unconditional -- a lane past the chunk's end reads the chunk's first row -- so that
     no branch stands between them. */
  const bool anyt = (ntied | nclosed) != 0u;
  for (u32 j0 = cs; j0 < ce; j0 += 128u) {
    u32 jj[2], v[2], tj[2], hj[2], head[2], th[2], ge[2], by[2];
    bool ok[2];
#pragma unroll
    for (u32 q = 0; q < 2u; q++) {
      const u32 j = j0 + 64u * q + lane;
      ok[q] = j < ce;
      jj[q] = ok[q] ? j : cs;
      v[q] = B->vA[jj[q]]; tj[q] = B->tied[jj[q]]; hj[q] = B->gh[jj[q]];
    }
#pragma unroll
    for (u32 q = 0; q < 2u; q++) {
      head[q] = (anyt && tj[q]) ? hj[q] : jj[q];
      th[q] = B->tied[head[q]]; ge[q] = B->gend[head[q]]; by[q] = S->inv[v[q] >> 24];
    }
#pragma unroll
    for (u32 q = 0; q < 2u; q++) {
      const u32 j = jj[q], idx = v[q] & 0x00FFFFFFu;
      const bool tdall = ok[q] && anyt && tj[q];                          /* tied: in the suffix array */
      const bool td = tdall && ntied && (th[q] & 2u);                      /* ... and of an open run: listed */
      if (ok[q]) {
        bwt[lo + j] = (u8)by[q];
        s.sa[lo + j] = SA_ENTRY(idx, v[q] >> 24) | ((tdall && head[q] != j) ? TIE_FLAG : 0u);
        if (idx == 0u) meta->bwt_idx = lo + j;
      }
      const u64 mask = __ballot(td);
      nlong += (u32)__popcll(__ballot(td && ge[q] - head[q] > BIG_RUN));
      if (td) {
        const u32 o = lbase + (u32)__popcll(mask & lanes_below());
        ls.sufx[o] = SA_ENTRY(idx, v[q] >> 24);
        ls.grp[o] = lo + head[q];                        /* rank of the run = its first row */
        ls.pos[o] = depth;                               /* symbols the run shares */
      }
      lbase += (u32)__popcll(mask);
    }
  }
  if (ntied && lane == 0u) { atomicMin(&S->lmin, depth); if (nlong) atomicAdd(&S->bc[9], nlong); }
  if (nclosed && lane == 0u) atomicMin(&S->cmin, depth);
  BT_MARK(tc2);
  BT_ADD(S, 4, tc1, tc2);
  if (lane == 0u) {
    const u64 tw2 = wall_clock64();
    atomicAdd(&S->bc[13], (u32)(tw2 - tw0));
    atomicAdd(&S->bc[14], (u32)(tw1 - tw0));
    atomicMax(&S->bc[5], (u32)(tw1 - tw0));
    atomicMax(&S->bc[6], (u32)(tw2 - tw0));
  }
}

/* Runs of rows whose keys agree after `>> sh`: fills gh (first row of the run), tied (run longer
 * than one row) and gend (indexed by first row).  Returns via *maxrun the longest run and the
 * number of tied rows of this thread.                                                      */
__device__ u32 batch_runs(batch_lds *B, const u64 *kR, u32 cnt, u32 sh, u32 *maxrun, bwt_lds *S)
{
  const u32 j0 = threadIdx.x * SORT_IPT;
  u32 headmask = 0, lasthead = 0, ntied = 0;
#pragma unroll
  for (u32 i = 0; i < SORT_IPT; i++) {
    const u32 j = j0 + i;
    if (j < cnt && (j == 0u || (kR[j] >> sh) != (kR[j - 1u] >> sh))) { headmask |= 1u << i; lasthead = j + 1u; }
  }
  u32 eh, d0, th, d1;
  wg_excl_max_add(lasthead, 0u, &eh, &d0, &th, &d1, &S->sc);
  u32 h1 = eh, longest = 0;
#pragma unroll
  for (u32 i = 0; i < SORT_IPT; i++) {
    const u32 j = j0 + i;
    if (j < cnt) {
      const bool hd = (headmask >> i) & 1u;
      if (hd) h1 = j + 1u;
      const bool hn = (j + 1u >= cnt) || ((kR[j + 1u] >> sh) != (kR[j] >> sh));
      const bool td = !(hd && hn);
      B->gh[j] = (u16)(h1 - 1u);
      B->tied[j] = td ? 1 : 0;
      ntied += td;
      if (hn) { B->gend[h1 - 1u] = (u16)(j + 1u); longest = (j + 2u - h1) > longest ? (j + 2u - h1) : longest; }
    }
  }
  *maxrun = wg_max(longest, &S->sc);
  return ntied;
}

/* Chunks of a batch whose runs (gh, gend) are known: chunk k = the groups that start inside
 * window k of CHUNK_WIN rows, [cstart[k], cstart[k+1]).  The bounds are fixed here, before any
 * wave starts rewriting the run tables; corder lists the chunks longest first (the longest
 * bounds the batch).  Waves then claim chunks with wave_claim().  Returns the number of windows. */
__device__ u32 chunk_plan(batch_lds *B, u32 cnt)
{
  const u32 tid = threadIdx.x;
  const u32 nwin = (cnt + CHUNK_WIN - 1u) / CHUNK_WIN;
  if (tid <= nwin) {
    const u32 w0 = tid * CHUNK_WIN;
    B->cstart[tid] = (u16)(w0 >= cnt ? cnt : ((B->gh[w0] == w0) ? w0 : B->gend[B->gh[w0]]));
  }
  __syncthreads();
  if (tid < nwin) {
    const u32 mine = (u32)B->cstart[tid + 1u] - (u32)B->cstart[tid];
    u32 r = 0;
    for (u32 j = 0; j < nwin; j++) {
      const u32 o = (u32)B->cstart[j + 1u] - (u32)B->cstart[j];
      r += (o > mine || (o == mine && j < tid)) ? 1u : 0u;
    }
    B->corder[r] = (u8)tid;
  }
  __syncthreads();
  return nwin;
}

/* Order rows [lo, lo+cnt) completely (as far as REFINE_ROUNDS reach) and emit them.
 * presorted: rows already sorted by the full key (cut at key boundaries); otherwise they are
 * only grouped by the partition's top MSD_BITS.  preloaded: the batch already sits in (kA,vA),
 * in text order (a whole small block).  Sets S->bc[8] if ties are left over.
 * Workgroup-wide steps: load, (rarely) a full LDS radix sort, the group scan.  Everything else
 * happens per wave on the groups that start in the wave's 256-row window.                   */
/* trim: rows [lo, lo+cnt) are as many as fit; a last group that continues beyond them is left
 * for the next batch (the caller advances by the return value; 0 = a single group fills the whole
 * batch).  Finding the cut on the rows already in LDS saves the separate search in HBM.      */
__device__ u32 batch_process(const u8 *T, u32 n, u8 *bwt, lbz_block_meta *meta, bwt_slot s,
                             bwt_lds *S, keycfg c, u32 lo, u32 cnt, bool presorted, bool preloaded,
                             bool trim = false, u32 depth = 0u,        /* depth: symbols the rows' keys reach (0: the first key, c.sy) */
                             const u64 *kfull = nullptr)               /* the rows' whole keys, where big_group has made them; else they are read off the text */
{
  batch_lds *B = &S->u.B;
  const u32 tid = threadIdx.x;
  const u64 tb0 = wall_clock64();
  /* the key behind the batch (the trim rule) and the batch's rows: every load of a thread is requested before the first
     is waited for -- as a loop of load, wait, store the four rows of a thread were four round trips to HBM, a tenth of
     the kernel (2.4 of 27.8 ms per block) */
  u64 behind = 0;
  const bool more = trim && lo + cnt < S->seghi;     /* partition depth <= 32 bits; the segment ends at a group boundary (beyond it a neighbour may be rewriting keys) */
  if (tid == 0 && more) behind = (u64)ldg_u32(reinterpret_cast<const u32 *>(s.k0) + lo + cnt) << 32;      /* (the partition's keys are top halves) */
  if (!preloaded) {
    u64 kk[BATCH_CAP / LBZ_WG];
    u32 vv[BATCH_CAP / LBZ_WG];
#pragma unroll
    for (u32 k = 0; k < BATCH_CAP / LBZ_WG; k++) {
      const u32 i0 = tid + k * LBZ_WG, i = i0 < cnt ? i0 : cnt - 1u;       /* cnt >= 1 */
      vv[k] = ldg_u32(s.v0 + lo + i);
      if (kfull) kk[k] = ldg_u64(kfull + lo + i);
      else kk[k] = (u64)ldg_u32(reinterpret_cast<const u32 *>(s.k0) + lo + i);
    }
    /* The partition moved the keys' top halves only (8-byte rows, round 6); the low half is four bytes of the text behind the
       top half's symbols (row_key): a 4-byte gather from the text of the four blocks an XCD works on at a time, in its L2. */
    if (!kfull) {
      /* all four gathers in flight before the first is waited for: the load is unconditional, from a place that lies inside the
         text (n > BATCH_ROWS here), and only a row whose four bytes wrap round the block's end (three rows of a block) takes the
         byte loop afterwards.  As a call of row_key per row each gather sat behind a branch and was waited for on its own: four
         round trips a batch, most of the load phase (2.9 us of a batch's 37) */
      u32 at[BATCH_CAP / LBZ_WG], raw[BATCH_CAP / LBZ_WG];
#pragma unroll
      for (u32 k = 0; k < BATCH_CAP / LBZ_WG; k++) {
        u32 a = (vv[k] & 0x00FFFFFFu) + c.q0;
        if (a >= n) a -= n;
        at[k] = a;
        raw[k] = ldg_text4(T + (a + 4u <= n ? a : n - 4u));
      }
#pragma unroll
      for (u32 k = 0; k < BATCH_CAP / LBZ_WG; k++) {
        u32 r = __builtin_bswap32(raw[k]);
        if (at[k] + 4u > n) r = (u32)row_key(T, n, 0u, vv[k] & 0x00FFFFFFu, c);
        kk[k] = (kk[k] << 32) | r;
      }
    }
#pragma unroll
    for (u32 k = 0; k < BATCH_CAP / LBZ_WG; k++) {
      const u32 i = tid + k * LBZ_WG;
      if (i < cnt) { B->kA[i] = kk[k]; B->vA[i] = vv[k]; }
    }
  }
  if (tid == 0) {
    S->bc[7] = 0; S->bc[5] = 0; S->bc[6] = 0;        /* window claim counter; the barriers below publish it */
    if (trim) S->bc[1] = more ? (u32)(behind >> S->msd_shift) : 0xFFFFFFFFu;
  }
  if (!preloaded) __syncthreads();
  const u64 tb1 = wall_clock64();
  bool need_sort = !presorted;
  u32 maxrun;
  if (preloaded) {
    if (lds_radix_sort(B, cnt, S)) {
      for (u32 i = tid; i < cnt; i += LBZ_WG) { B->kA[i] = B->kB[i]; B->vA[i] = B->vB[i]; }
      __syncthreads();
    }
    need_sort = false;
  }
  batch_runs(B, B->kA, cnt, need_sort ? S->msd_shift : 0u, &maxrun, S);
  BT_MARK(tg1);
  if (threadIdx.x == 0u) { BT_ADD(S, 5, tb1, tg1); }
  if (trim && lo + cnt < S->seghi && (u32)(B->kA[cnt - 1u] >> S->msd_shift) == S->bc[1]) {
    cnt = B->gh[cnt - 1u];                       /* the last group goes on: it waits for the next batch */
    if (cnt == 0u) { __syncthreads(); return 0u; }
  }
  if (need_sort && maxrun > WAVE_GROUP) {
#ifdef LDS_SORT_TICKS
    const u64 tl0 = wall_clock64();
#endif
    if (lds_radix_sort(B, cnt, S)) {
      for (u32 i = tid; i < cnt; i += LBZ_WG) { B->kA[i] = B->kB[i]; B->vA[i] = B->vB[i]; }
      __syncthreads();
    }
    need_sort = false;
    batch_runs(B, B->kA, cnt, 0u, &maxrun, S);
#ifdef LDS_SORT_TICKS
    if (tid == 0) { S->dbg[1] += (u32)(wall_clock64() - tl0); S->dbg[0] += 1u; }
#endif
  }
  if (need_sort && maxrun > SPLIT_MIN) {
    /* A long group would be one wave's job from start to end and hold the whole batch up.  Cut
       every such group into sub-groups first: one wave per group orders it on its most
       significant varying key byte (one counting pass) and marks the runs of that byte as
       groups, which the chunking below then deals out over several waves.               */
    if (tid == 0) S->bc[2] = 0;
    __syncthreads();
#pragma unroll
    for (u32 i = 0; i < SORT_IPT; i++) {
      const u32 j = tid * SORT_IPT + i;
      if (j < cnt && B->gh[j] == j && (u32)B->gend[j] - j > SPLIT_MIN) B->cstart[atomicAdd(&S->bc[2], 1u)] = (u16)j;
    }
    __syncthreads();
    const u32 nlong = S->bc[2];                               /* <= BATCH_CAP / SPLIT_MIN < size of cstart */
    for (u32 i = wave_id(); i < nlong; i += LBZ_NW) {
      const u32 gs = B->cstart[i], ge = B->gend[gs];
      const u32 sh = wave_radix_range<true>(B, gs, ge);
      if (sh < 64u) wave_runs<false>(B, gs, ge, sh);
    }
    __syncthreads();
  }
  const u64 tb2 = wall_clock64();
  if (threadIdx.x == 0u) { BT_ADD(S, 6, tg1, tb2); }
  const u32 nwin = chunk_plan(B, cnt);
  BT_MARK(tg3);
  if (threadIdx.x == 0u) { BT_ADD(S, 7, tb2, tg3); BT_ADD(S, 9, 0, 1); BT_ADD(S, 10, 0, cnt); }
  for (;;) {
    const u32 t = wave_claim(&S->bc[7]);
    if (t >= nwin) break;
    const u32 k = B->corder[t];
    const u32 cs = B->cstart[k], ce = B->cstart[k + 1u];
    if (cs < ce) wave_finish_chunk(B, cs, ce, need_sort, depth ? depth : c.sy, bwt, s, lo, meta, S);
  }
  __syncthreads();
  if (tid == 0) {
    S->bc[10] += (u32)(tb1 - tb0); S->bc[11] += (u32)(tb2 - tb1); S->bc[12] += (u32)(wall_clock64() - tb2);
    S->bc[3] += S->bc[5]; S->bc[4] += S->bc[6];          /* longest single chunk: first sort, everything */
  }
  return cnt;
}

/* is the run of equal keys in rows [lo,hi) CLOSED (wave_finish_chunk): the same byte in front of every row, rotation 0 not among them? */
__device__ bool rows_closed(bwt_slot s, bwt_lds *S, u32 lo, u32 hi)
{
  const u32 c0 = s.v0[lo] >> 24;
  u32 bad = 0;
  for (u32 j = lo + threadIdx.x; j < hi; j += LBZ_WG) {
    const u32 v = s.v0[j];
    if ((v >> 24) != c0 || (v & 0x00FFFFFFu) == 0u) bad = 1u;
  }
  return wg_max(bad, &S->sc) == 0u;
}

/* rows [lo,hi) share one 64-bit key and are more than a batch: left to the rank rounds -- or, a closed run, as they are */
__device__ void emit_tied_rows(u8 *bwt, bwt_slot s, bwt_lds *S, u32 lo, u32 hi, lbz_block_meta *meta, u32 depth, bool closed = false)
{
  for (u32 j = lo + threadIdx.x; j < hi; j += LBZ_WG) {
    const u32 v = s.v0[j];
    bwt[j] = S->inv[v >> 24];
    s.sa[j] = SA_ENTRY(v & 0x00FFFFFFu, v >> 24) | (j > lo ? TIE_FLAG : 0u);
    if ((v & 0x00FFFFFFu) == 0u) meta->bwt_idx = j;
  }
  if (threadIdx.x == 0) {
    if (closed) atomicMin(&S->cmin, depth);
    else { S->bc[8] = 1u; atomicMin(&S->h0min, depth); }
  }
#ifdef DEEP_DEBUG
  if (threadIdx.x == 0) printf("emit_tied_rows [%u,%u) depth %u\n", lo, hi, depth);
#endif
  __syncthreads();
}

/* last q in (pos, e] with a key change after `>> sh` between rows q-1 and q; 0 if none */
template <class K>
__device__ u32 find_cut(const K *keys, u32 pos, u32 e, u32 sh, bwt_lds *S)
{
  const u32 tid = threadIdx.x;
  u32 found = 0;
  for (u32 back = 0; back < BATCH_CAP && !found; back += LBZ_WG) {
    u32 cand = 0;
    if (e >= back + tid) {
      const u32 q = e - back - tid;
      if (q > pos && (keys[q] >> sh) != (keys[q - 1u] >> sh)) cand = q;
    }
    found = wg_max(cand, &S->sc);
  }
  return found;
}

/* first q in [from, hi) whose key differs from row pos after `>> sh`; hi if none.  The keys are sorted on
 * the bits above sh, so this is a search for a boundary: LBZ_WG probes spread evenly over the range narrow
 * it to 1/LBZ_WG per step (three steps for a whole block) instead of a scan of LBZ_WG rows per step.   */
template <class K>
__device__ u32 find_run_end(const K *keys, u32 pos, u32 from, u32 hi, u32 sh, bwt_lds *S)
{
  const K k = keys[pos] >> sh;
  u32 a = from, b = hi;                         /* invariant: rows < a belong to the run, row b (or hi) does not */
  while (a < b) {
    const u32 span = b - a;
    const u32 step = (span + LBZ_WG - 1u) / LBZ_WG;           /* probe i looks at row a + i * step */
    const u32 q = a + threadIdx.x * step;
    const u32 cand = (q < b && (keys[q] >> sh) != k) ? threadIdx.x : 0xFFFFFFFFu;
    const u32 f = wg_min(cand, &S->sc);                       /* first probe outside the run */
    if (f == 0xFFFFFFFFu) {                                   /* every probe inside: the boundary is behind the last probe */
      const u32 last = a + ((span - 1u) / step) * step;
      a = last + 1u;
    } else {
      b = a + f * step;
      a = f ? a + (f - 1u) * step + 1u : a;
      if (f == 0u) { b = a; }
    }
    if (step == 1u) break;
  }
  return b < hi ? b : hi;
}

/* rows [lo,hi) share one 64-bit key and are more than a batch, but few enough for one wave of k_bwt_deep to take apart
 * symbol by symbol (deep_big_run): they join the segment's list as one run */
#ifndef LONG_RUN_MAX
#define LONG_RUN_MAX 4096u
#endif
__device__ void list_tied_rows(u8 *bwt, bwt_slot s, bwt_lds *S, u32 lo, u32 hi, lbz_block_meta *meta, u32 depth)
{
  if (threadIdx.x == 0) { S->bc[1] = atomicAdd(&S->listn, hi - lo); atomicMin(&S->lmin, depth); S->bc[9] += hi - lo; }
  __syncthreads();
  const u32 base = S->bc[1];
  const bwt_slot ls = seg_view(s, S->seglo);
  for (u32 j = lo + threadIdx.x; j < hi; j += LBZ_WG) {
    const u32 v = s.v0[j];
    const u32 e = SA_ENTRY(v & 0x00FFFFFFu, v >> 24);
    bwt[j] = S->inv[v >> 24];
    s.sa[j] = e | (j > lo ? TIE_FLAG : 0u);
    if ((v & 0x00FFFFFFu) == 0u) meta->bwt_idx = j;
    ls.sufx[base + (j - lo)] = e; ls.grp[base + (j - lo)] = lo; ls.pos[base + (j - lo)] = depth;
  }
  __syncthreads();
}

/* An oversized group [lo,hi) (> BATCH_CAP rows with equal top MSD_BITS): HBM radix sort on the remaining key bits, then
 * batches cut at key boundaries.  A run of more than a batch of EQUAL keys (" of the ": thousands of rows of a text
 * block) goes to the text rounds' list as it is, one long run for one wave to take apart -- up to LONG_RUN_MAX rows.  A longer
 * one (eight blanks: a third of the rows of a block of indented sources; the zero padding of a tar) is the whole workgroup's
 * job (round 6): its rows get NEW keys -- the next sy symbols of the text -- are sorted on them in HBM and walked again, one
 * level deeper, at most BIG_LEVELS times; the levels are frames on a small stack in LDS (where a frame ends, how deep its keys
 * reach): keys of different frames are of different depths, so nothing is compared across a frame's end.  What is still one
 * run of more than LONG_RUN_MAX rows then ("abababab": periodic stretches) is left to the rank rounds, and the block with it.
 * Until round 6 every such run sent its block there -- with the closed runs that is a tenfold detour (every block of a real
 * tar of sources: profiles/r06_a_rows_realtar.txt).                                                                        */
#ifndef BIG_LEVELS
#define BIG_LEVELS 12u
#endif
__device__ void big_group(const u8 *T, u32 n, u8 *bwt, lbz_block_meta *meta, bwt_slot s,
                          bwt_lds *S, keycfg c, u32 lo, u32 hi)
{
  const u32 tid = threadIdx.x;
  const u32 m = hi - lo;
  /* whole keys for the group's rows (the partition's are top halves): in the second key column, sorted with the rank table's
     column as the other buffer -- both idle until the text rounds */
  u64 *const K = s.k1, *const K2 = s.isa;
  for (u32 j = lo + tid; j < hi; j += LBZ_WG) K[j] = row_key(T, n, reinterpret_cast<const u32 *>(s.k0)[j], s.v0[j] & 0x00FFFFFFu, c);
  __syncthreads();
  const u32 which = wg_radix_sort(K + lo, s.v0 + lo, K2 + lo, s.v1 + lo, m, S->msd_shift, S);
  if (which) {
    for (u32 i = tid; i < m; i += LBZ_WG) { K[lo + i] = K2[lo + i]; s.v0[lo + i] = s.v1[lo + i]; }
  }
  if (tid == 0) { S->fr_end[0] = hi; S->fr_dep[0] = BATCH_DEPTH(c); }
  __syncthreads();
  u32 pos = lo, nf = 1u;                                /* frames in use (every thread keeps the same count) */
  while (pos < hi) {
    while (pos >= S->fr_end[nf - 1u]) nf--;             /* (frame 0 ends at hi) */
    const u32 lim = S->fr_end[nf - 1u], dep = S->fr_dep[nf - 1u];
    u32 e = pos + BATCH_ROWS < lim ? pos + BATCH_ROWS : lim;
    if (e < lim) {
      const u32 cut = find_cut(K, pos, e, 0u, S);
      if (!cut) {
        const u32 end = find_run_end(K, pos, e, lim, 0u, S);
#ifndef DBG_NOCLOSE_EMIT
        if (rows_closed(s, S, pos, end)) emit_tied_rows(bwt, s, S, pos, end, meta, dep, true);
        else
#endif
        if (end - pos <= LONG_RUN_MAX) list_tied_rows(bwt, s, S, pos, end, meta, dep);
        else if (nf <= BIG_LEVELS && dep + c.sy < n) {
          const u32 g = end - pos;
          for (u32 j = pos + tid; j < end; j += LBZ_WG) {
            u32 start = (s.v0[j] & 0x00FFFFFFu) + dep;
            if (start >= n) start -= n;
            K[j] = key_from_text(T, n, start, S->cmap, c);
          }
          __syncthreads();
          const u32 w2 = wg_radix_sort(K + pos, s.v0 + pos, K2 + pos, s.v1 + pos, g, 64u, S);
          if (w2) {
            for (u32 i = tid; i < g; i += LBZ_WG) { K[pos + i] = K2[pos + i]; s.v0[pos + i] = s.v1[pos + i]; }
          }
          if (tid == 0) { S->fr_end[nf] = end; S->fr_dep[nf] = dep + c.sy; }
          __syncthreads();
          nf++;
          continue;                                     /* the same rows again, by their new keys */
        }
        else emit_tied_rows(bwt, s, S, pos, end, meta, dep);
        pos = end;
        continue;
      }
      e = cut;
    }
    batch_process(T, n, bwt, meta, s, S, c, pos, e - pos, true, false, false, dep, K);
    pos = e;
  }
}

/* ======================================================================= deep ties
 * Prefix doubling on a suffix array whose unresolved rows carry TIE_FLAG.  State: isa[] (rank
 * of every rotation = first row of its run) and the list of still-tied rows in row order
 * (sufx, grp = rank, pos = row).  One round at depth h re-sorts every run by the rank of the
 * rotation h symbols further on and splits it where those ranks differ.  Runs are short and
 * many, so rounds work like k_bwt_batch: consecutive whole runs of <= BATCH_CAP rows are pulled
 * into LDS (the rank lookups are the only random HBM reads), each wave orders the runs of its
 * window, and rows leave the list as soon as they are unique.  A run longer than a batch goes
 * through the HBM radix sorter.                                                            */
__device__ u32 doubling_round(const u8 *T, u32 n, u8 *bwt, bwt_slot s, bwt_lds *S, u32 h, u32 m, u32 tag)
{
  batch_lds *B = &S->u.B;
  const u32 tid = threadIdx.x, lane = lane_id();
  u32 out_m = 0, k0 = 0;
  while (k0 < m) {
    u32 e = k0 + BATCH_ROWS < m ? k0 + BATCH_ROWS : m;
    if (e < m) {
      const u32 cut = find_cut(s.grp, k0, e, 0u, S);
      if (!cut) {
        /* a run of more than BATCH_CAP rows: sort it by the looked-up rank in HBM, then split */
        const u32 end = find_run_end(s.grp, k0, e, m, 0u, S);
        const u32 g = end - k0, rowbase = s.pos[k0];
        __syncthreads();
        for (u32 i = tid; i < g; i += LBZ_WG) {
          const u32 sf = s.sufx[k0 + i];
          u32 t = SA_IDX(sf) + h;
          if (t >= n) t -= n;
          s.k0[i] = (u64)isa_before(s.isa[t], tag);
          s.v0[i] = sf;
        }
        __syncthreads();
        const u32 which = wg_radix_sort(s.k0, s.v0, s.k1, s.v1, g, RANK_BITS, S);
        out_m = wg_regroup<false>(which ? s.k1 : s.k0, which ? s.v1 : s.v0, rowbase, out_m, g, s, S, T, n, bwt, 0xFFFFFFFFu, s.grp[k0], tag);
        k0 = end;
        continue;
      }
      e = cut;
    }
    const u32 cnt = e - k0;
    const u64 td0 = wall_clock64();
    if (tid == 0) { S->bc[7] = 0; S->bc[6] = 0; }        /* chunk tickets of the two phases */
    for (u32 i = tid; i < cnt; i += LBZ_WG) {
      const u32 sf = s.sufx[k0 + i];
      u32 t = SA_IDX(sf) + h;
      if (t >= n) t -= n;
      B->kA[i] = (u64)isa_before(s.isa[t], tag);
      B->vA[i] = sf;
      B->kB[i] = (u64)s.grp[k0 + i];
    }
    __syncthreads();
    const u64 td1 = wall_clock64();
    u32 maxrun;
    batch_runs(B, B->kB, cnt, 0u, &maxrun, S);            /* the runs as they stand */
    const u64 td2 = wall_clock64();
    /* sort phase: waves claim chunks, longest first (as in batch_process) */
    const u32 nwin = chunk_plan(B, cnt);
    for (;;) {
      const u32 t = wave_claim(&S->bc[7]);
      if (t >= nwin) break;
      const u32 k = B->corder[t];
      const u32 cs = B->cstart[k], ce = B->cstart[k + 1u];
      u32 mytied = 0;
      if (cs < ce) {
        wave_sort_chunk<true>(B, cs, ce);       /* ranks are 20-bit keys: (key << 12 | row) loses nothing */
        /* rows of the sorted positions (a run occupies consecutive rows), kept in kB */
        for (u32 j = cs + lane; j < ce; j += 64u) {
          const u32 gs = B->gh[j];
          B->kB[j] = (u64)(s.pos[k0 + gs] + (j - gs));
          B->ghn[j] = (u16)gs;                  /* first position of the run as it stood: its row is the rank so far */
        }
        wave_sync();
        mytied = wave_runs<true>(B, cs, ce);
      }
      if (lane == 0u) B->ctied[k] = (u16)mytied;
    }
    __syncthreads();
    const u64 td3 = wall_clock64();
    /* write phase: a chunk's still-tied rows go behind those of the chunks before it */
    const u32 myct = lane < nwin ? (u32)B->ctied[lane] : 0u;         /* nwin <= 64: one lane per chunk */
    const u32 total = wave_sum(myct);
    for (;;) {
      const u32 k = wave_claim(&S->bc[6]);
      if (k >= nwin) break;
      const u32 cs = B->cstart[k], ce = B->cstart[k + 1u];
      u32 off = out_m + wave_sum(lane < k ? myct : 0u);
      for (u32 j0 = cs; j0 < ce; j0 += 64u) {
        const u32 j = j0 + lane;
        const bool ok = j < ce;
        const bool td = ok && B->tied[j];
        const u64 mask = __ballot(td);
        if (ok) {
          const u32 row = (u32)B->kB[j], sf = B->vA[j];
          const u32 newrank = (u32)B->kB[B->gh[j]];
          /* the suffix array itself is not read again: ranks (isa) and the list carry the rounds, the
             BWT byte of a row that became unique is written below */
          /* the first piece of a run that splits keeps the run's rank: no rank store for its rows (a third of the
             scattered stores of a block with deep ties) */
          const u32 oh = B->ghn[j];
          if ((u32)B->gh[j] != oh) s.isa[SA_IDX(sf)] = ISA_ENTRY(newrank, (u32)B->kB[oh], tag);
          if (td) {
            const u32 o = off + (u32)__popcll(mask & lanes_below());
            s.sufx[o] = sf; s.grp[o] = newrank; s.pos[o] = row;
          } else {
            bwt[row] = S->inv[SA_CODE(sf)];
          }
        }
        off += (u32)__popcll(mask);
      }
    }
    __syncthreads();
    if (tid == 0) {
      S->bc[10] += (u32)(td1 - td0); S->bc[11] += (u32)(td2 - td1); S->bc[12] += (u32)(td3 - td2);
      S->bc[13] += (u32)(wall_clock64() - td3); S->bc[14] += 1u;
    }
    out_m += total;
    k0 = e;
  }
  return out_m;
}

/* dense symbol codes of the used bytes and the key geometry they allow (every kernel) */
__device__ keycfg bwt_setup(const lbz_block_meta *meta, bwt_lds *S)
{
  const u32 tid = threadIdx.x;
  u32 ninuse;
  const u32 f = (tid < 256u && meta->inuse[tid]) ? 1u : 0u;
  const u32 ex = wg_excl_add(f, &ninuse, &S->sc);
  /* dense codes pack more symbols into a key; with more than 128 byte values in use they are 8 bits like the bytes, and the
     bytes serve as their own codes (keys are then the text itself: key_from_text, k_bwt_deep) */
  if (tid < 256u) S->cmap[tid] = ninuse > 128u ? (u8)tid : (u8)ex;
  if (ninuse > 128u) { if (tid < 256u) S->inv[tid] = (u8)tid; }
  else if (f) S->inv[ex] = (u8)tid;
  keycfg c;
  c.b = 1u;
  while ((1u << c.b) < ninuse) c.b++;
  c.sy = 64u / c.b;
  if (c.sy > MAX_SYMS) c.sy = MAX_SYMS;
  c.pad = 64u - c.b * c.sy;
  c.q0 = 32u / c.b;
  if (tid == 0) {
    for (u32 i = 1; i < 16; i++) S->bc[i] = 0;
    S->msd_shift = 64u - (meta->msd_bits ? meta->msd_bits : MSD_BITS);
  }
  __syncthreads();
  return c;
}


/* segments of a block (see "segments" below): how many, and where one begins */
__device__ __forceinline__ u32 bwt_nseg(u32 n, u32 segs)          /* segs: the launch's workgroups per block (<= LBZ_BWT_MAXSEGS) */
{
  const u32 p = n / (4u * SMALL_BLOCK);             /* SMALL_BLOCK = a batch of k_bwt_batch, in either build of this file */
  return p < 1u ? 1u : (p > segs ? segs : p);
}

/* first row at or behind x that starts a group of the partition; n if there is none */
__device__ u32 seg_cut(const u64 *k0, u32 x, u32 n, bwt_lds *S)
{
  if (x == 0u) return 0u;
  if (x >= n) return n;
  return find_run_end(reinterpret_cast<const u32 *>(k0), x - 1u, x, n, S->msd_shift - 32u, S);      /* the partition's keys: top halves */
}

/* ---- kernels 1: the partition, several workgroups per block ----------------------------------------------------------
 * Stable least-significant-digit-first radix passes on the key's top 32 (16) bits, as before -- but a block's rows are dealt
 * over up to PART_MAX workgroups: workgroup j of a pass takes the j-th RANGE of the pass's input order (ranges of 2^k rows,
 * whole tiles) and scatters it in order, starting for every digit behind the rows of that digit in the ranges before it.
 * What it needs for that -- the digit histogram of every range of its pass -- is counted by the pass BEFORE it while it
 * scatters: a row's place in the output says which range of the next pass it will be read in, its next digit is in the
 * key, so the histogram of the next pass grows in LDS ([range][digit], 16 KB) and is added to the block's table in HBM
 * when the workgroup is done.  One launch per pass, no workgroup waits for another; the kernel boundary is the
 * synchronisation.  k_bwt_hist counts for the first pass (from the text), k_bwt_segs fixes the segments at the end.
 * Until round 4 one workgroup took a block through all its passes: 14 ms per block whatever the device was doing, two
 * waves of workgroups for 1112 blocks on 1024 slots, and the whole of a small input's latency.                          */
struct part_lds {
  wg_scratch sc;
  u32 bc[16];
  u32 listn, seglo;
  u32 h0min, lmin;
  u32 cmin, cpad;
  u32 fr_end[BIG_FRAMES], fr_dep[BIG_FRAMES];
  u32 msd_shift, seghi;
  u32 dbg[4];
  u8 cmap[256];
  u8 inv[256];
  sort_lds X;
  u32 nh[PART_MAX][256];                /* digit histogram of the NEXT pass, per range of its input (= this pass's output rows) */
};
static_assert(offsetof(part_lds, X) == offsetof(bwt_lds, u), "same layout up to the union");
static_assert(sizeof(part_lds) <= 54 * 1024, "three per CU");
struct scat_lds {                       /* k_bwt_scat: the same without the digit histograms of `X` (it reads its offsets from the block's tables) */
  wg_scratch sc;
  u32 bc[16];
  u32 listn, seglo;
  u32 h0min, lmin;
  u32 cmin, cpad;
  u32 fr_end[BIG_FRAMES], fr_dep[BIG_FRAMES];
  u32 msd_shift, seghi;
  u32 dbg[4];
  u8 cmap[256];
  u8 inv[256];
  sort_core X;
  u32 nh[PART_MAX][256];
};
static_assert(offsetof(scat_lds, X) == offsetof(bwt_lds, u), "same layout up to the union");
static_assert(sizeof(scat_lds) <= 40 * 1024, "four per CU");

/* ranges of 2^shift rows (whole tiles), at most `parts` of them */
__device__ __forceinline__ u32 part_shift(u32 n, u32 parts)
{
  u32 sh = 10u;                                  /* SORT_TILE = 1024 rows in the main build */
  while (((n + (1u << sh) - 1u) >> sh) > parts) sh++;
  return sh;
}
/* the block's four histogram tables: [table][range][digit] */
__device__ __forceinline__ u32 *part_table(bwt_slot s, u32 t) { return s.ph + (size_t)t * PART_MAX * 256u; }

/* (block of the round, part) of this workgroup: the parts of a block on one XCD (seg_item's dealing) */
__device__ __forceinline__ bool part_item(u32 nblk, u32 parts, u32 *i, u32 *j)
{
  const u32 g = blockIdx.x, k = g >> 3;
  *i = (k / parts) * 8u + (g & 7u);
  *j = k % parts;
  return *i < nblk;
}

/* the first pass's digits, counted per range of the text: tables 0 (32-bit partition: the key byte at bit 32) and 1 (16-bit:
   at bit 48); also the sorter's part of the block record */
__global__ void __launch_bounds__(LBZ_WG, 3)
k_bwt_hist(const u8 *Tbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 parts,
           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs)
{
  __shared__ part_lds S_;
  bwt_lds &S = *reinterpret_cast<bwt_lds *>(&S_);
  const u32 tid = threadIdx.x;
  u32 bi, j;
  if (!part_item(nblk, parts, &bi, &j)) return;
  const u32 blk = lbz_round_block(first, count, bi, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (j == 0u && tid == 0) {                  /* the segment workgroups add to these */
    M->periodic = 0; M->rounds = 0; M->sort_elems = 0; M->deep_rows = 0; M->nseg = 0; M->deep_h0 = 0xFFFFFFFFu; M->deep_closed = 0xFFFFFFFFu; M->deep_skip = 0; M->deep_long = 0;
    for (u32 i = 0; i <= LBZ_DEEP_ROUNDS; i++) { M->deep_tot[i] = 0; M->deep_hmin[i] = 0xFFFFFFFFu; }
    M->msd_bits = MSD_BITS;
    for (u32 i = 0; i < 8u; i++) M->ticks[i] = 0;
    for (u32 i = 0; i < 16u; i++) M->fticks[i] = 0;
    for (u32 i = 0; i < LBZ_BWT_MAXSEGS; i++) M->seg_m[i] = 0;
    if (n <= SMALL_BLOCK) { M->nseg = 1u; M->seg_lo[0] = 0u; M->seg_lo[1] = n; }     /* small blocks are sorted whole by k_bwt_batch */
  }
  if (n <= SMALL_BLOCK) return;
  const u32 sh = part_shift(n, parts);
  const u32 r0 = j << sh;
  if (r0 >= n) return;
  const u32 r1 = r0 + (1u << sh) < n ? r0 + (1u << sh) : n;
  const bwt_slot s = round_slot(ws, slot_bytes, ws_spill, spill_bytes, count, L, bi);
  const u8 *T = Tbase + lbz_elem_off(L, blk);
  const keycfg c = bwt_setup(M, &S);
  sort_lds *P = &S.u.X;
  for (u32 i = tid; i < 4u * 256u; i += LBZ_WG) (&P->hist[0][0])[i] = 0;
  __syncthreads();
  msd_text_pass<false>(T, n, c, nullptr, nullptr, &S, 0u, r0, r1);
  if (tid < 256u) {
    part_table(s, 0)[j * 256u + tid] = P->hist[0][tid];
    part_table(s, 1)[j * 256u + tid] = P->hist[2][tid];
    part_table(s, 2)[j * 256u + tid] = 0u;                    /* the first pass adds the second pass's counts here */
  }
}

/* pass `pass` of the block's 2 or 4: range j of the pass's input */
__global__ void __launch_bounds__(LBZ_WG, 4)
k_bwt_scat(const u8 *Tbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 parts,
           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 pass)
{
  __shared__ scat_lds S_;
  bwt_lds &S = *reinterpret_cast<bwt_lds *>(&S_);
  const u32 tid = threadIdx.x;
  u32 bi, j;
  if (!part_item(nblk, parts, &bi, &j)) return;
  const u32 blk = lbz_round_block(first, count, bi, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (n <= SMALL_BLOCK) return;
  const u32 sh = part_shift(n, parts);
  const u32 nr = (n + (1u << sh) - 1u) >> sh;                  /* ranges in use */
  const u32 r0 = j << sh;
  if (r0 >= n) return;
  const u32 r1 = r0 + (1u << sh) < n ? r0 + (1u << sh) : n;
  const u64 tk0 = wall_clock64();
  const bwt_slot s = round_slot(ws, slot_bytes, ws_spill, spill_bytes, count, L, bi);
  const u8 *T = Tbase + lbz_elem_off(L, blk);
  const keycfg c = bwt_setup(M, &S);
  sort_lds *P = &S.u.X;
  /* How deep to partition (every workgroup of the block works it out the same way from the first pass's table, part 0
     notes it): with 8-bit symbols the digit histogram is the text's histogram, and no byte value in more than 1/128 of
     the positions (incompressible data) -> 16 bits, its groups are a dozen rows then; else MSD_BITS.                 */
  u32 bits = M->msd_bits;                                      /* passes behind the first: what the first one noted (table 0 is in use again) */
  if (pass == 0u) {
    u32 tot0 = 0;
    if (tid < 256u) for (u32 i = 0; i < nr; i++) tot0 += part_table(s, 0)[i * 256u + tid];
    const u32 top = wg_max(tot0, &S.sc);
    bits = (c.b == 8u && MSD_BITS > MSD_BITS_FLAT && top <= n / 128u) ? MSD_BITS_FLAT : MSD_BITS;
  }
  const u32 passes = bits / 8u;
  if (pass >= passes) return;
  if (pass == 0u && j == 0u && tid == 0) M->msd_bits = bits;
  const u32 dj = 4u - passes + pass;                           /* digits 4 - passes .. 3, least significant first */
  /* tables: the first pass reads 0 or 1; from then on pass p reads what pass p - 1 added up, adds to the next and clears
     the one after that (its own range's rows) for the pass after it */
  const u32 tin = pass == 0u ? (bits == MSD_BITS ? 0u : 1u) : (pass == 1u ? 2u : (pass == 2u ? 3u : 0u));
  const u32 tout = pass == 0u ? 2u : (pass == 1u ? 3u : 0u);
  const u32 tzero = pass == 0u ? 3u : (pass == 1u ? 0u : 1u);
  const bool next = pass + 1u < passes;
  {
    const u32 *H = part_table(s, tin);
    u32 tot = 0, before = 0;
    if (tid < 256u)
      for (u32 i = 0; i < nr; i++) {
        const u32 h = H[i * 256u + tid];
        if (i < j) before += h;
        tot += h;
      }
    u32 all;
    const u32 ex = wg_excl_add(tot, &all, &S.sc);
    if (tid < 256u) P->dbase[tid] = ex + before;
    for (u32 i = tid; i < PART_MAX * 256u; i += LBZ_WG) (&S_.nh[0][0])[i] = 0;
    if (tid < 256u && pass + 2u < passes) part_table(s, tzero)[j * 256u + tid] = 0u;
    __syncthreads();
  }
  /* buffers alternate so that the last pass lands in (k0,v0) */
  u32 *kb[2] = { reinterpret_cast<u32 *>(s.k0), reinterpret_cast<u32 *>(s.k1) };      /* rows of 8 bytes: the key's top half and the value */
  u32 *vb[2] = { s.v0, s.v1 };
  const u32 dst = (passes - 1u - pass) & 1u;                  /* pass p writes buffer (passes - 1 - p) & 1 */
  const u32 shift = 32u + 8u * dj;
  if (pass == 0u) {
    if (next) msd_text_pass<true, true>(T, n, c, kb[dst], vb[dst], &S, shift, r0, r1, S_.nh, sh);
    else msd_text_pass<true, false>(T, n, c, kb[dst], vb[dst], &S, shift, r0, r1);
  } else {
    if (next) msd_array_pass<true>(kb[dst ^ 1u], vb[dst ^ 1u], shift, kb[dst], vb[dst], &S, r0, r1, S_.nh, sh);
    else msd_array_pass<false>(kb[dst ^ 1u], vb[dst ^ 1u], shift, kb[dst], vb[dst], &S, r0, r1, nullptr, 0u);
  }
  if (next) {
    __syncthreads();
    u32 *O = part_table(s, tout);
    for (u32 i = tid; i < nr * 256u; i += LBZ_WG) {
      const u32 v = (&S_.nh[0][0])[i];
      if (v) atomicAdd(&O[i], v);
    }
  }
  if (tid == 0) atomicAdd(&M->ticks[2], (u32)(wall_clock64() - tk0));
}

/* The partition as ONE workgroup per block, all passes in one launch (the only form until round 4).  Where rounds of
 * several hundred blocks overlap on several streams it is still the better one -- a long kernel of few waves that the other
 * streams' kernels fill in around (7.97 against 7.5-7.8 GB/s for the launch-per-pass form on wiki(10^9), three streams,
 * same box) -- while a single round, or a round of few blocks, waits for a block's 14 ms of passes; lbz_api.hip picks.  */
struct part1_lds {
  wg_scratch sc;
  u32 bc[16];
  u32 listn, seglo;
  u32 h0min, lmin;
  u32 cmin, cpad;
  u32 fr_end[BIG_FRAMES], fr_dep[BIG_FRAMES];
  u32 msd_shift, seghi;
  u32 dbg[4];
  u8 cmap[256];
  u8 inv[256];
  sort_lds X;
  u32 nh[1][256];                       /* digit histogram of the next pass (symbols narrower than a byte: counted while scattering) */
};
static_assert(offsetof(part1_lds, X) == offsetof(bwt_lds, u), "same layout up to the union");
__global__ void __launch_bounds__(LBZ_WG, 4)
k_bwt_part(const u8 *Tbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count,
           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 segs)
{
  __shared__ part1_lds S_;
  bwt_lds &S = *reinterpret_cast<bwt_lds *>(&S_);
  const u32 tid = threadIdx.x;
  const u32 blk = lbz_round_block(first, count, blockIdx.x, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (tid == 0) {                             /* the sorter's part of the block record: the segment workgroups add to it */
    M->periodic = 0; M->rounds = 0; M->sort_elems = 0; M->deep_rows = 0; M->nseg = 0; M->deep_h0 = 0xFFFFFFFFu; M->deep_closed = 0xFFFFFFFFu; M->deep_skip = 0; M->deep_long = 0;
    for (u32 i = 0; i <= LBZ_DEEP_ROUNDS; i++) { M->deep_tot[i] = 0; M->deep_hmin[i] = 0xFFFFFFFFu; }
    M->msd_bits = MSD_BITS;
    for (u32 i = 0; i < 8u; i++) M->ticks[i] = 0;
    for (u32 i = 0; i < 16u; i++) M->fticks[i] = 0;
    for (u32 i = 0; i < LBZ_BWT_MAXSEGS; i++) M->seg_m[i] = 0;
    if (n <= SMALL_BLOCK) { M->nseg = 1u; M->seg_lo[0] = 0u; M->seg_lo[1] = n; }
  }
  if (n <= SMALL_BLOCK) return;               /* small blocks are sorted whole by k_bwt_batch */
  const bwt_slot s = round_slot(ws, slot_bytes, ws_spill, spill_bytes, count, L, blockIdx.x);
  const u8 *T = Tbase + lbz_elem_off(L, blk);
  const u64 tk0 = wall_clock64();
  const keycfg c = bwt_setup(M, &S);
  sort_lds *P = &S.u.X;
  for (u32 i = tid; i < 4u * 256u; i += LBZ_WG) (&P->hist[0][0])[i] = 0;
  __syncthreads();
  msd_text_pass<false>(T, n, c, nullptr, nullptr, &S, 0u, 0u, n);           /* hist[0], hist[2]: the key bytes at bits 32 and 48 */
  /* With 8-bit symbols a key byte is one symbol of the rotation, and every position of the text is the k-th symbol of
     exactly one rotation: every digit's histogram is the text's, so none but the first is counted -- and it tells how
     deep to partition (k_bwt_scat).  Narrower symbols straddle the key bytes: a pass counts the next pass's digits.  */
  const bool bytes = c.b == 8u;
  const u32 top = wg_max(tid < 256u ? P->hist[0][tid] : 0u, &S.sc);
  const u32 bits = (bytes && MSD_BITS > MSD_BITS_FLAT && top <= n / 128u) ? MSD_BITS_FLAT : MSD_BITS;
  if (tid == 0) M->msd_bits = bits;
  const u32 passes = bits / 8u;
  u32 *kb[2] = { reinterpret_cast<u32 *>(s.k0), reinterpret_cast<u32 *>(s.k1) };
  u32 *vb[2] = { s.v0, s.v1 };
  load_digit_offsets(passes == 4u ? P->hist[0] : P->hist[2], P->dbase, &S);
  u32 cur = (passes - 1u) & 1u;                                /* buffers alternate so that the last pass lands in (k0,v0) */
  u32 (*nh)[256] = bytes ? nullptr : S_.nh;                   /* narrow symbols: a pass counts the next pass's digits */
  if (nh) { for (u32 i = tid; i < 256u; i += LBZ_WG) S_.nh[0][i] = 0; __syncthreads(); }
  if (nh && passes > 1u) msd_text_pass<true, true>(T, n, c, kb[cur], vb[cur], &S, 32u + 8u * (4u - passes), 0u, n, nh, 31u);
  else msd_text_pass<true, false>(T, n, c, kb[cur], vb[cur], &S, 32u + 8u * (4u - passes), 0u, n);
  for (u32 pass = 1; pass < passes; pass++) {
    __syncthreads();
    load_digit_offsets(bytes ? P->hist[0] : S_.nh[0], P->dbase, &S);
    const bool next = pass + 1u < passes && nh;
    if (next) { for (u32 i = tid; i < 256u; i += LBZ_WG) S_.nh[0][i] = 0; __syncthreads(); }
    if (next) msd_array_pass<true>(kb[cur], vb[cur], 32u + 8u * (4u - passes + pass), kb[cur ^ 1u], vb[cur ^ 1u], &S, 0u, n, nh, 31u);
    else msd_array_pass<false>(kb[cur], vb[cur], 32u + 8u * (4u - passes + pass), kb[cur ^ 1u], vb[cur ^ 1u], &S, 0u, n, nullptr, 0u);
    cur ^= 1u;
  }
  /* the segments' bounds (k_bwt_segs) */
  __syncthreads();
  if (tid == 0) S.msd_shift = 64u - bits;
  __syncthreads();
  const u32 nseg = bwt_nseg(n, segs);
  for (u32 g = 1; g < nseg; g++) {
    const u32 cut = seg_cut(s.k0, (u32)((u64)g * n / nseg), n, &S);
    if (tid == 0) M->seg_lo[g] = cut;
  }
  if (tid == 0) { M->nseg = nseg; M->seg_lo[0] = 0u; M->seg_lo[nseg] = n; M->ticks[2] = (u32)(wall_clock64() - tk0); }
}

/* The segments' bounds: group boundaries of the partition nearest to the even cuts.  Fixed here, before k_bwt_batch: its
   segment workgroups must not search the key column of a neighbour at work.  One workgroup per block.               */
struct segs_lds {                       /* bwt_lds up to its union: all the boundary searches touch */
  wg_scratch sc;
  u32 bc[16];
  u32 listn, seglo;
  u32 h0min, lmin;
  u32 cmin, cpad;
  u32 fr_end[BIG_FRAMES], fr_dep[BIG_FRAMES];
  u32 msd_shift, seghi;
  u32 dbg[4];
  u8 cmap[256];
  u8 inv[256];
};
static_assert(sizeof(segs_lds) == offsetof(bwt_lds, u), "same layout up to the union");
__global__ void __launch_bounds__(LBZ_WG, 4)
k_bwt_segs(lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 segs,
           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs)
{
  __shared__ segs_lds S_;
  bwt_lds &S = *reinterpret_cast<bwt_lds *>(&S_);
  const u32 tid = threadIdx.x;
  const u32 blk = lbz_round_block(first, count, blockIdx.x, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (n <= SMALL_BLOCK) return;
  const bwt_slot s = round_slot(ws, slot_bytes, ws_spill, spill_bytes, count, L, blockIdx.x);
  if (tid == 0) S.msd_shift = 64u - M->msd_bits;
  __syncthreads();
#ifdef DEEP_DEBUG
  if (tid == 0) {
    u32 bad = 0;
    for (u32 i = 1; i < n && bad < 5u; i++)
      if ((reinterpret_cast<const u32 *>(s.k0)[i] >> (S.msd_shift - 32u)) < (reinterpret_cast<const u32 *>(s.k0)[i - 1u] >> (S.msd_shift - 32u))) { printf("part: blk %u row %u out of order (bits %u)\n", blk, i, M->msd_bits); bad++; }
    u64 sum = 0; for (u32 i = 0; i < n; i++) sum += s.v0[i] & 0xFFFFFFu;
    printf("part: blk %u n %u index sum %llu (want %llu)\n", blk, n, sum, (u64)n * (n - 1u) / 2u);
  }
#endif
  const u32 nseg = bwt_nseg(n, segs);
  for (u32 g = 1; g < nseg; g++) {
    const u32 cut = seg_cut(s.k0, (u32)((u64)g * n / nseg), n, &S);
    if (tid == 0) M->seg_lo[g] = cut;
  }
  if (tid == 0) { M->nseg = nseg; M->seg_lo[0] = 0u; M->seg_lo[nseg] = n; }
}

/* ---- segments -------------------------------------------------------------------------------------------
 * After the partition the rows of a block are grouped by their top MSD_BITS and the groups are independent of
 * each other, so the rest of the sort does not need one workgroup per block: the rows are cut into up to
 * LBZ_BWT_SEGS segments at group boundaries and every (block, segment) is a workgroup of its own -- in
 * k_bwt_batch and in each launch of the deep-tie rounds.  Eight times as many, eight times shorter work items:
 * a launch no longer ends on a device that is a fifth full (1112 blocks on 512 resident workgroups were three
 * waves of workgroups, the last one 17 % full), an input of a hundred blocks fills the chip, and a block's
 * chain of stages is short enough for the work-unit interface.  The segment workgroups of a block share its
 * workspace slot; what they share beyond that is isa[]: see k_bwt_fixr.                                       */
/* (block of the round, segment) of this workgroup.  Hardware deals workgroup j to XCD j mod 8: the segment
 * workgroups of one block sit on ONE XCD (they share the block's text and ranks in that L2) and within 64
 * positions of each other in dispatch order.  Grid = ceil(nblk / 8) * 8 * segs.                                 */
__device__ __forceinline__ bool seg_item(u32 nblk, u32 segs, u32 *i, u32 *seg)
{
  const u32 j = blockIdx.x, k = j >> 3;
  *i = (k / segs) * 8u + (j & 7u);
  *seg = k % segs;
  return *i < nblk;
}

/* ---- kernel 2: LDS batches of whole groups; emits BWT bytes + rows; flags deep ties ---- */
#ifndef BATCH_WGS
#define BATCH_WGS 5                     /* workgroups of k_bwt_batch a CU is to hold: what the registers are budgeted for (LDS: BATCH_ROWS) */
#endif
__global__ void __launch_bounds__(LBZ_WG, BATCH_WGS)
k_bwt_batch(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 segs,
            u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs)
{
  __shared__ bwt_lds S;
  const u32 tid = threadIdx.x;
  u32 bi, seg;
  if (!seg_item(nblk, segs, &bi, &seg)) return;
  const u32 blk = lbz_round_block(first, count, bi, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (n == 0u) return;
  const u32 nseg = n == 1u ? 1u : M->nseg;    /* k_bwt_part fixed the segments (blocks of one byte never reach it: n == 1 below) */
  if (seg >= nseg) return;
  const bwt_slot s = round_slot(ws, slot_bytes, ws_spill, spill_bytes, count, L, bi);
  const size_t off = lbz_elem_off(L, blk);
  const u8 *T = Tbase + off;
  u8 *bwt = Bbase + off;
  if (n == 1u) {                              /* divbwt.c:1712 */
    if (tid == 0) { bwt[0] = T[0]; M->bwt_idx = 0; M->periodic = 0; M->nseg = 1; M->seg_lo[0] = 0; M->seg_lo[1] = 1; }
    return;
  }
  const u64 tk0 = wall_clock64();
  const keycfg c = bwt_setup(M, &S);
  u32 lo = 0, hi = n;
  if (nseg > 1u) { lo = M->seg_lo[seg]; hi = M->seg_lo[seg + 1u]; }
  if (tid == 0) {
    S.listn = 0; S.seglo = lo; S.seghi = hi; S.h0min = 0xFFFFFFFFu; S.lmin = 0xFFFFFFFFu; S.cmin = 0xFFFFFFFFu;
    for (u32 i = 0; i < 4; i++) S.dbg[i] = 0;
#ifdef BATCH_TICKS
    for (u32 i = 0; i < 16; i++) S.u.B.bt[i] = 0;
#endif
  }
  __syncthreads();
  if (n <= BATCH_ROWS) {
    batch_lds *B = &S.u.B;
    for (u32 i = tid; i < n; i += LBZ_WG) {
      B->kA[i] = key_from_text(T, n, i, S.cmap, c);
      B->vA[i] = ((u32)S.cmap[T[i ? i - 1u : n - 1u]] << 24) | i;
    }
    __syncthreads();
    batch_process(T, n, bwt, M, s, &S, c, 0u, n, false, true);
  } else {
    u32 pos = lo;
    while (pos < hi) {
      const u32 want = hi - pos < BATCH_ROWS ? hi - pos : BATCH_ROWS;
      const u32 used = batch_process(T, n, bwt, M, s, &S, c, pos, want, false, false, true, BATCH_DEPTH(c));
      if (used == 0u) {                          /* one group fills the batch: sort it in HBM first */
#ifdef LDS_SORT_TICKS
        const u64 tg0 = wall_clock64();
#endif
        const u32 end = find_run_end(reinterpret_cast<const u32 *>(s.k0), pos, pos + want, hi, S.msd_shift - 32u, &S);   /* inside the segment (the partition's keys: top halves) */
        big_group(T, n, bwt, M, s, &S, c, pos, end);
#ifdef LDS_SORT_TICKS
        if (tid == 0) { S.dbg[3] += (u32)(wall_clock64() - tg0); S.dbg[2] += end - pos; }
#endif
        pos = end;
        continue;
      }
      pos += used;
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (S.bc[8]) { atomicMax(&M->periodic, LBZ_TIES_EARLY); atomicMin(&M->deep_h0, S.h0min); M->deep_skip = 1u; }  /* ties left for the rank rounds: long runs BIG_ROUNDS did not split */
    M->seg_m[seg] = S.listn;                   /* short runs: the text rounds' list */
    if (S.listn) { atomicAdd(&M->deep_tot[0], S.listn); atomicMin(&M->deep_hmin[0], S.lmin); }
    if (S.bc[9]) atomicAdd(&M->deep_long, S.bc[9]);
    if (S.cmin != 0xFFFFFFFFu) atomicMin(&M->deep_closed, S.cmin);
    atomicAdd(&M->deep_rows, S.listn);
    atomicAdd(&M->sort_elems, hi - lo);
    /* diagnostics, summed over the block's segments (tests/tools/quickperf.py) */
    atomicAdd(&M->ticks[0], (u32)(wall_clock64() - tk0));
#ifdef BATCH_TICKS
    for (u32 i = 0; i < 16; i++) atomicAdd(&M->fticks[i], S.u.B.bt[i]);
#endif
#ifndef DEEP_TICKS                                 /* (that diagnostic build keeps ticks[5..7] for the long runs of the text rounds) */
    for (u32 i = 0; i < 3; i++) atomicAdd(&M->ticks[3 + i], S.bc[10 + i]);   /* load, group scan (+block sorts), per-wave part */
#ifndef COL_TICKS
    atomicAdd(&M->ticks[6], S.bc[13]); atomicAdd(&M->ticks[7], S.bc[14]);
#endif
#endif
#ifdef LDS_SORT_TICKS
    atomicAdd(&M->ticks[1], S.dbg[0]); atomicAdd(&M->ticks[2], S.dbg[1]);
#else
    atomicAdd(&M->ticks[1], S.bc[3]); atomicAdd(&M->ticks[2], S.bc[4]);    /* summed over waves: busy, of which first sort */
#endif
  }
}

/* ---- kernel 2b: the text rounds -- short runs of tied rows, ordered a 64-lane strip at a time ----
 * k_bwt_batch leaves the rows that are tied in runs of at most BIG_RUN rows (on text two thirds of a block: equal on
 * their first 8 symbols, rarely on their first 40) in a list per segment: (suffix + code of the byte before it, rank =
 * first row of the run, symbols the run is known to share), the rows of a run side by side.  What orders them is the
 * text itself: 0.9 MB per block that the segment workgroups of a block keep in their XCD's L2, read-only -- against
 * 7.2 MB of rank entries per block that the rank rounds gather from and scatter into across HBM (64 useful bits per
 * 128-byte line, at the chip's 50 G random accesses a second: profiles/r04_micro_random.json).
 *
 * A wave claims DEEP_CHUNK list entries at a time and owns the runs that START inside; it walks them in strips of
 * whole runs, one row per lane.  A step: every tied lane loads the 16 bytes behind the symbols its run shares (one
 * unaligned global_load_dwordx4, big-endian = string order) and the strip is sorted inside its runs on 2 x 52 bits of
 * them (deep_stage): the key (first lane of the run : 52 bits : lane) is unique, so a row's place in the strip is the
 * number of smaller keys -- counted against all 64 keys, which sit in LDS and are read at a wave-uniform address
 * (broadcast, two per ds_read_b128): one compare and one add per pair, no branch, no dependence on run lengths.  The
 * rows move through LDS to their places, runs split where neighbours differ.  Steps repeat while enough of the strip
 * is tied (a strip that thins out is left to the next launch, which gets the survivors side by side again); launch r
 * allows more steps per strip than the one before, so long repeats -- pairs of rows as a rule -- are followed for
 * thousands of symbols by a strip that holds nothing else.  Rows that become unique write their BWT byte; every row
 * writes the suffix array (the rank rounds rebuild their lists from its tie flags should they be needed).
 * Whatever is still tied after DEEP_ROUNDS launches (exactly periodic blocks, repeats of more than 6 KB) marks the
 * block for the rank rounds, which start at the least depth such a run has reached (deep_h0).                     */
struct u64x2 { u64 x, y; };
#ifndef DEEP_NEAR
#define DEEP_NEAR 16u                   /* a strip whose longest run has at most this many rows counts a row's place among the keys of its own run (8 or 16) */
#endif
struct deep_wave {
  alignas(16) u64 comp[64 + DEEP_NEAR];   /* the strip's keys, read by every lane; the tail stays ~0 */
  u64 srt[64];                          /* (run, slice) in sorted order */
  u64 k2[64];                           /* the second slice travels with its row */
  u32 val[64];
  u32 cnt[256], fill[256];              /* long runs: rows per value of the symbol they are split on, rows placed so far */
  u16 base[256], obase[256];            /* ... first row of each value's rows inside the run, and inside the run's stretch of the list */
  u32 st_off[DEEP_STACK], st_len[DEEP_STACK], st_dep[DEEP_STACK], st_buf[DEEP_STACK], sp;   /* pieces of a long run that are still long */
};
struct deep_lds {
  wg_scratch sc;
  u32 ticket, outn, h0min, bad, cmin;
  u8 inv[256];
  deep_wave w[LBZ_NW];
};

/* steps a strip may take in launch r: 2 2 4 8 16 32 256 256.  A text step decides 13 symbols, or skips 16 to 64 that all
   its runs share; a rank step (from launch DEEP_BUILD + 1 on) as many as the run it looks up shares */
#ifndef DEEP_K0
#define DEEP_K0 2u                      /* text steps a strip may take in the first launch ... */
#endif
#ifndef DEEP_K1
#define DEEP_K1 2u                      /* ... and in the second */
#endif
__device__ __forceinline__ u32 deep_kmax(u32 round) { return round == 0u ? DEEP_K0 : (round == 1u ? DEEP_K1 : (round + 2u < DEEP_ROUNDS ? 2u << (round - 1u) : 256u)); }

/* Sort the strip inside its runs on `slice` (52 bits).  val and k2 move with their rows; hl (first lane of the lane's
 * run) and tied describe places, and are refined.  Lanes >= nv are not part of the strip.
 * A row's place is the number of smaller keys, and the key begins with the run's first lane: every key of a run in front of
 * mine is smaller, every key of a run behind it greater.  So only the keys of my OWN run need counting -- the rows from its
 * first lane on, as many as the strip's longest run has (round 5: DEEP_NEAR) -- and the rest is the run's first lane itself.
 * On text nine strips in ten hold no run of more than 16 rows: 2 to 16 compare-and-add pairs a row and slice at a per-lane
 * LDS address instead of 64 at a broadcast one (the count was a third of the kernel's vector instructions: DESIGN 3.2, 4).
 * A strip with a longer run counts against all 64 keys as before. */
__device__ __forceinline__ u32 deep_longest_run(u32 hl, u32 lane);
template <u32 N>
__device__ __forceinline__ u32 deep_count_near(const u64 *cp, u32 pos, u64 comp)
{
  constexpr u32 G = N < 16u ? N : 16u;
#pragma unroll
  for (u32 q0 = 0; q0 < N; q0 += G) {
    u64 c[G];
#pragma unroll
    for (u32 q = 0; q < G; q++) c[q] = cp[q0 + q];
#pragma unroll
    for (u32 q = 0; q < G; q += 2u) pos = add_if_less2(pos, c[q], c[q + 1u], comp);
  }
  return pos;
}
__device__ __forceinline__ void deep_stage(deep_wave *W, u32 lane, u32 nv, u64 slice, u32 &val, u64 &k2, u32 &hl, bool &tied)
{
  const bool in = lane < nv;
  const u64 ck = ((u64)hl << 52) | slice;
  const u64 comp = (ck << 6) | (u64)lane;
  W->comp[lane] = in ? comp : ~0ull;
#ifndef DEEP_ALLPAIRS
  /* the strip's longest run, from the first lanes of its runs (lanes >= nv are runs of one) */
  const u32 lmax = deep_longest_run(hl, lane);
#endif
  wave_sync();
  u32 pos = 0;
#ifndef DEEP_ALLPAIRS
  if (lmax <= DEEP_NEAR) {
    const u64 *cp = W->comp + hl;                          /* comp[64 .. 64 + DEEP_NEAR) hold ~0: a run at the strip's end reads on */
    if (lmax <= 2u) pos = deep_count_near<2>(cp, hl, comp);
    else if (lmax <= 4u) pos = deep_count_near<4>(cp, hl, comp);
    else if (lmax <= 8u) pos = deep_count_near<8>(cp, hl, comp);
    else pos = deep_count_near<16>(cp, hl, comp);
  } else
#endif
  {
    const u64x2 *cp = reinterpret_cast<const u64x2 *>(W->comp);
    /* 16 keys at a time: eight broadcast reads in flight, then sixteen compare-and-add pairs */
#pragma unroll
    for (u32 b = 0; b < 4u; b++) {
      if (16u * b < nv) {
        u64x2 c2[8];
#pragma unroll
        for (u32 q = 0; q < 8u; q++) c2[q] = cp[8u * b + q];
#pragma unroll
        for (u32 q = 0; q < 8u; q++) pos = add_if_less2(pos, c2[q].x, c2[q].y, comp);
      }
    }
  }
  if (in) { W->srt[pos] = ck; W->val[pos] = val; W->k2[pos] = k2; }
  wave_sync();
  u64 me = 0, below = 1;
  if (in) {
    val = W->val[lane]; k2 = W->k2[lane];
    me = W->srt[lane];
    below = lane ? W->srt[lane - 1u] : ~me;
  }
  const bool head = !in || me != below;
  hl = wave_incl_max(head ? lane : 0u);
  const u64 hm = __ballot(head);
  const bool nexthead = lane == 63u || ((hm >> (lane + 1u)) & 1ull);
  tied = in && !(head && nexthead);
  wave_sync();
}

/* The strip's longest run (wave-uniform; anything above DEEP_NEAR counts as "long"). */
__device__ __forceinline__ u32 deep_longest_run(u32 hl, u32 lane)
{
  const u64 hm0 = __ballot(hl == lane);
  const u64 ab = lane == 63u ? 0ull : hm0 >> (lane + 1u);
  const u32 he = ab ? lane + 1u + (u32)__builtin_ctzll(ab) : 64u;
  return wave_max(hl == lane ? he - lane : 0u);
}

/* CLOSED RUNS in a strip (see wave_finish_chunk): a run whose rows all have the same byte in front of them, rotation 0 not among
 * them, is finished whatever order its rows are in.  Called when the strip's runs have just split: the rows of such a run stop
 * being `tied` (they take no further steps and are not listed again) and are closed (their output keeps them tied in the
 * suffix array, for the rank rounds should the block come to need them: a row that is not tied but not alone in its run).  Runs
 * only split, and every part of a closed run is closed, so the answer is taken afresh from the rows as they stand. */
__device__ __forceinline__ void deep_close(u32 val, u32 hl, u32 lane, bool in, bool &tied)
{
  const u32 code = SA_CODE(val);
  const u32 hcode = (u32)__shfl((int)code, (int)hl);                  /* the byte in front of the run's first row */
  const u64 bad = __ballot(in && (code != hcode || SA_IDX(val) == 0u));
  const u64 hm = __ballot(hl == lane);                                 /* first lanes of the runs as they stand */
  const u64 above = lane == 63u ? 0ull : hm >> (lane + 1u);
  const u32 he = above ? lane + 1u + (u32)__builtin_ctzll(above) : 64u;
  const u64 runmask = (he == 64u ? ~0ull : (1ull << he) - 1ull) & ~((1ull << hl) - 1ull);
#ifdef DBG_NOCLOSE_STRIP
  const bool open = true;
#else
  const bool open = (bad & runmask) != 0ull;
#endif
  tied = tied && open;
}

/* text of rotation idx from symbol d on: 16 bytes, memory order (first symbol in the low byte of .a) */
__device__ __forceinline__ u64x2 deep_load16(const u8 *T, u32 n, u32 idx, u32 d)
{
  u32 at = idx + d;
  if (at >= n) at -= n;
  if (at + 16u <= n) {
    const lbz_text16 *q = reinterpret_cast<const lbz_text16 *>(T + at);
    u64x2 r; r.x = q->a; r.y = q->b;
    return r;
  }
  u64 xa = 0, xb = 0;
  for (u32 q = 0; q < 16u; q++) {
    const u64 by = T[at];
    if (q < 8u) xa |= by << (8u * q); else xb |= by << (8u * (q - 8u));
    at = at + 1u == n ? 0u : at + 1u;
  }
  u64x2 x; x.x = xa; x.y = xb;
  return x;
}

/* A strip of few rows in a long repeat (the late launches: pairs of passages that occur twice, thousands of symbols deep -- and
 * since the closed runs (deep_close) the rows such a pair would look up have no rank entries: their order cannot show, so nobody
 * ranked them).  The idle lanes look ahead: lane = (chunk, row), every row's next 64 / cp chunks of 16 bytes are compared with
 * the same chunk of its run's first row, and all rows advance by the chunks every run agrees on -- a pair takes 512 bytes a trip
 * instead of 64.  Returns the symbols to advance by (for the lanes that take text steps; 0 for the others). */
__device__ __forceinline__ u32 deep_wide_skip(const u8 *T, u32 n, u32 val, u32 hl, bool useT, u32 d, u32 cut, u32 lane)
{
  const u32 cp = cut <= 2u ? 2u : (cut <= 4u ? 4u : (cut <= 8u ? 8u : (cut <= 16u ? 16u : 32u)));
  const u32 r = lane & (cp - 1u), cix = lane / cp;
  const u32 rv = (u32)__shfl((int)val, (int)r), rh = (u32)__shfl((int)hl, (int)r);
  const bool ru = __shfl((int)(useT ? 1 : 0), (int)r) != 0;
  u32 rd = (u32)__shfl((int)d, (int)r), total = 0;
  for (u32 it = 0; it < 8192u; it++) {
    const bool valid = ru && rd + 16u * (cix + 1u) <= n;
    u64x2 x; x.x = 0; x.y = 0;
    if (valid) x = deep_load16(T, n, SA_IDX(rv), rd + 16u * cix);
    const u32 from = rh + cix * cp;                          /* the same chunk of the run's first row */
    const u64 ha = (u64)(u32)__shfl((int)(u32)x.x, (int)from) | (u64)(u32)__shfl((int)(u32)(x.x >> 32), (int)from) << 32;
    const u64 hb = (u64)(u32)__shfl((int)(u32)x.y, (int)from) | (u64)(u32)__shfl((int)(u32)(x.y >> 32), (int)from) << 32;
    const u64 dm = __ballot(ru && (!valid || x.x != ha || x.y != hb));
    const u32 adv = dm ? (u32)__builtin_ctzll(dm) / cp : 64u / cp;      /* chunks every run agrees on (lanes are chunk-major) */
    rd += 16u * adv;
    total += 16u * adv;
    if (dm) break;
  }
  return useT ? total : 0u;
}

/* Ranks for the rows the text has not ordered by launch DEEP_BUILD (on text a quarter of a block: repeats of 35 symbols and
 * more).  What is left then is mostly LONG repeats -- passages that occur twice -- and walking those symbol by symbol costs
 * their length squared; a run x, y tied for d symbols is ordered by the ranks of x + d and y + d instead (prefix doubling).
 * The rank rounds (k_bwt_fix*) keep a rank for EVERY rotation, 7 MB per block written and gathered across HBM; here only
 * the rows in the lists have entries (isa[], same 8-byte {rank, rank before, tag} words, so that a launch reads the ranks
 * as they stood when it began whatever its other workgroups are writing), and a bit map of the block's rotations says
 * which: a target without an entry became unique before launch DEEP_BUILD ended, so the text decides within the few dozen
 * symbols those launches covered, and the strip takes a text step instead.  From then on every change of rank is stored.  */
/* The invariant the concurrent segment workgroups of a block rely on (round-4 review): a rank entry is ONE naturally aligned
 * 64-bit word, read and written by single 8-byte global accesses (global_load_dwordx2 / global_store_dwordx2 are single-copy
 * atomic on aligned addresses), so a reader sees a whole entry of some launch, never halves of two.  In the text rounds an
 * entry may be written TWICE in a launch -- a tag-0 refresh {rank, rank, depth} when its strip starts, a tagged update
 * {new rank, old rank, tag, depth} when the strip ends -- and a reader of either takes the rank as it stood when the launch
 * began: the refresh carries it in both fields, the update in `rank before` (isa_before with the launch's tag).  The depth
 * note is a lower bound in both.  The rank rounds (k_bwt_fixr) write an entry at most once per launch. */
struct deep_ranks { u64 *isa; u32 *map; u32 tag, hcur; bool build, live; };
__device__ __forceinline__ void deep_publish(deep_ranks R, u32 idx, u32 rank, u32 depth)
{
  R.isa[idx] = ISA_ENTRY_D(rank, rank, 0u, depth);
  atomicOr(&R.map[idx >> 5], 1u << (idx & 31u));
}

struct deep_lists { const u32 *sin, *gin, *din; u32 *sout, *gout, *dout; };

/* A run of g >= 64 tied rows (list entries [p, p + g)): too long for a strip.  Its wave takes it apart symbol by symbol:
 * it finds how many further symbols ALL rows of the piece share (compared with the piece's first row, 16 bytes a step, up
 * to 64), then splits the piece on the first symbol they do not all share -- a counting sort on one byte whose only state
 * in LDS is 256 counters: a row's slot inside its value's rows is the return value of an LDS atomic (the order inside a
 * sub-run is free, it is still tied).  Rows alone with their value are done; values with 2..63 rows become runs of the next
 * list (the strips of the next launch order them); values with 64 rows or more are pieces again -- their suffixes go to the
 * other of two scratch columns (the run's own stretch of the incoming list and the partition's value column, free by now)
 * and onto a small stack.  " of the ", four thousand rows of a text block, is thirty pieces after one symbol and short runs
 * after two or three; a template that a hundred rows share for 60 symbols costs four 16-byte steps.  A piece that shares
 * 64 further symbols (or overflows the stack) goes to the next list as it is, deeper, and the next launch carries on.
 * (Round 5, measured: 2 600 pieces and 490 000 piece-rows per text block, 43 % of a text launch's wave time, 3-5 us for
 * each of a piece's four phases whatever its size -- profiles/r05_deep_ticks.txt.  Ordering pieces of up to 256 rows in one
 * pass instead -- 16 bytes a row fetched once, a 56-bit key of the seven symbols behind the shared ones, all-pairs count
 * over the piece's keys in LDS -- halves the long runs' wave time (46.9 -> 26.3 ms per block) and costs 24 vector registers:
 * at five waves a SIMD the tie stages gain 2-4 % on text and lose 2 % on sources, at four they lose 8 %.  So k_bwt_deep
 * runs this without it (MID = false); k_bwt_long, the long runs in a launch of their own, with it.)                          */
__device__ void deep_big_run(deep_wave *W, u32 lane, u32 p, u32 g, deep_lists Ls, u32 *ping, u32 *pong, const u8 *T, u32 n,
                             u32 *sa, u8 *bwt, const u8 *inv, lbz_block_meta *M, u32 *outn, u32 &hmin, u32 *cminp, bool late, deep_ranks R)
{
  const u32 rank0 = Ls.gin[p];
  const u32 d0 = Ls.din[p];
  /* how far a piece is followed in one launch: 64 shared symbols and DEEP_LEVELS splits in the first launches (a long run must not keep
     its wave while the strips of a hundred thousand rows wait); in the last two, whose lists are short, as far as it takes -- seven
     hundred files of one block that begin with the same licence text are ONE run for six hundred symbols, a table sheds a few
     rows with every symbol for a hundred (profiles/r06_b_rows_realtar.txt: such blocks ended in the rank rounds) */
  const u32 maxit = R.tag + 2u <= DEEP_ROUNDS ? (R.tag <= 2u ? 4u : 16u) : 2048u;
  const u32 maxlev = R.tag + 2u <= DEEP_ROUNDS ? DEEP_LEVELS : 40u;
  wave_sync();                                          /* (every lane has read the empty stack of the run before) */
  if (lane == 0u) { W->st_off[0] = 0u; W->st_len[0] = g; W->st_dep[0] = d0; W->st_buf[0] = 0u; W->sp = 1u; }
  if (R.live)                                                     /* as for the strips: rank and depth at the start of the launch */
    for (u32 k = lane; k < g; k += 64u) R.isa[SA_IDX(ping[p + k])] = ISA_ENTRY_D(rank0, rank0, 0u, d0);
  wave_sync();
  for (;;) {
    const u32 sp = W->sp;
    if (sp == 0u) break;
    const u32 off = W->st_off[sp - 1u], len = W->st_len[sp - 1u], buf = W->st_buf[sp - 1u] & 1u, level = W->st_buf[sp - 1u] >> 1;
    u32 d = W->st_dep[sp - 1u];
    wave_sync();
    if (lane == 0u) W->sp = sp - 1u;
    const u32 *src = (buf ? pong : ping) + p + off;
    u32 *dst = (buf ? ping : pong) + p + off;
    const u32 r0 = rank0 + off;                         /* the piece's rows are consecutive from here */
    const u32 idx0 = SA_IDX(src[0]);
    bool split = false;
#ifdef DEEP_TICKS
    const u64 tb0 = wall_clock64();
#endif
    for (u32 it = 0; it < maxit && d + 16u <= n; it++) {
      const u64x2 ref = deep_load16(T, n, idx0, d);
      u32 lc = 16u;
      for (u32 k0 = 0; k0 < len; k0 += 256u) {           /* four strips a trip: their loads are in flight together */
        u32 v4[4];
        u64x2 x4[4];
#pragma unroll
        for (u32 q = 0; q < 4u; q++) { const u32 k = k0 + 64u * q + lane; v4[q] = src[k < len ? k : 0u]; }
        /* unconditional loads from inside the text, the wrap round the block's end (sixteen rotations of a block) looked at
           afterwards: as four calls of deep_load16 each load sat behind that test's branch and was waited for on its own */
        u32 a4[4];
#pragma unroll
        for (u32 q = 0; q < 4u; q++) {
          u32 at = SA_IDX(v4[q]) + d;
          if (at >= n) at -= n;
          a4[q] = at;
          const lbz_text16 *t16 = reinterpret_cast<const lbz_text16 *>(T + (at + 16u <= n ? at : (n >= 16u ? n - 16u : 0u)));
          x4[q].x = t16->a; x4[q].y = t16->b;
        }
#pragma unroll
        for (u32 q = 0; q < 4u; q++) if (a4[q] + 16u > n) x4[q] = deep_load16(T, n, SA_IDX(v4[q]), d);
#pragma unroll
        for (u32 q = 0; q < 4u; q++) {
          const u64 xa = x4[q].x ^ ref.x, xb = x4[q].y ^ ref.y;
          const u32 l = xa ? (u32)__builtin_ctzll(xa) >> 3 : (xb ? 8u + ((u32)__builtin_ctzll(xb) >> 3) : 16u);
          lc = l < lc ? l : lc;
        }
      }
      lc = wave_min(lc);
      d += lc;
      if (lc < 16u) {
        split = true;
        break;
      }
    }
    if (!split || d >= n || level >= maxlev) {
      /* nothing to split on yet (64 more symbols shared, or tied all the way round), or enough for one launch (a run that
         sheds a few rows with every symbol -- counters, tables -- would keep its wave for as many passes as it is deep):
         the piece goes on as it is, deeper */
      const u32 ob = wave_reserve(outn, len);
      for (u32 k = lane; k < len; k += 64u) {
        const u32 val = src[k];
        Ls.sout[ob + k] = val; Ls.gout[ob + k] = r0; Ls.dout[ob + k] = d;
        if (late) {                                     /* the rows may have moved inside the run since the suffix array was written */
          sa[r0 + k] = val | (k ? TIE_FLAG : 0u);
          if (SA_IDX(val) == 0u) M->bwt_idx = r0 + k;
        }
        if (R.build) deep_publish(R, SA_IDX(val), r0, d);
        else if (R.live && (off || d != d0)) R.isa[SA_IDX(val)] = ISA_ENTRY_D(r0, rank0, R.tag, d);
      }
      hmin = d < hmin ? d : hmin;
      wave_sync();
      continue;
    }
#ifdef DEEP_TICKS
    const u64 tb1 = wall_clock64();
#endif
    for (u32 i = lane; i < 256u; i += 64u) { W->cnt[i] = 0; W->fill[i] = 0; }
    wave_sync();
    for (u32 k0 = 0; k0 < len; k0 += 256u) {
      u32 v4[4], b4[4];
      {
#pragma unroll
        for (u32 q = 0; q < 4u; q++) { const u32 k = k0 + 64u * q + lane; v4[q] = src[k < len ? k : 0u]; }
#pragma unroll
        for (u32 q = 0; q < 4u; q++) { u32 at = SA_IDX(v4[q]) + d; if (at >= n) at -= n; b4[q] = T[at]; }
      }
      /* closed sub-runs (deep_close): a value's rows all have the same byte in front of them iff the sum of those bytes' codes is
         rows x the largest code; rotation 0 counts as a code of its own.  The sum rides above the row count (13 bits: a run in a
         list has at most LONG_RUN_MAX = 4096 rows), exact up to 2048 rows -- a longer sub-run is taken for open */
#pragma unroll
      for (u32 q = 0; q < 4u; q++)
        if (k0 + 64u * q + lane < len) {
          const u32 code = SA_CODE(v4[q]);
          atomicAdd(&W->cnt[b4[q]], 1u | (code << 13));
          atomicMax(&W->fill[b4[q]], SA_IDX(v4[q]) == 0u ? 256u : code);
        }
    }
    wave_sync();
#ifdef DEEP_TICKS
    const u64 tb2 = wall_clock64();
#endif
    u32 shorttot;
    bool anyclosed = false;
    {
      u32 c[4], t[4];
      bool pushed[4], clo[4];
      u32 sum = 0, tsum = 0;
#pragma unroll
      for (u32 q = 0; q < 4u; q++) {
        const u32 cw = W->cnt[4u * lane + q], mx = W->fill[4u * lane + q];
        c[q] = cw & 0x1FFFu;
#ifdef DBG_NOCLOSE_BIG
        clo[q] = false;
#else
        clo[q] = c[q] > 1u && c[q] <= 2048u && mx < 256u && (cw >> 13) == c[q] * mx;
#endif
        anyclosed |= clo[q];
        pushed[q] = false;
        if (c[q] > BIG_RUN && !clo[q]) {                /* a piece again, if the stack has room: else it goes to the next list as a long run */
          const u32 at = atomicAdd(&W->sp, 1u);
          pushed[q] = at < DEEP_STACK;
          if (pushed[q]) { W->st_len[at] = c[q]; W->st_dep[at] = d + 1u; W->st_buf[at] = (buf ^ 1u) | ((level + 1u) << 1); W->st_off[at] = 4u * lane + q; /* the value: its offset follows */ }
        }
        t[q] = (c[q] > 1u && !pushed[q] && !clo[q]) ? c[q] : 0u;
        sum += c[q]; tsum += t[q];
      }
      wave_sync();                                     /* every lane has read its counters: they become plain counts (bit 31: closed), the slots start at zero */
#pragma unroll
      for (u32 q = 0; q < 4u; q++) { W->cnt[4u * lane + q] = c[q] | (clo[q] ? 0x80000000u : 0u); W->fill[4u * lane + q] = 0u; }
      const u32 tin = wave_incl_add(tsum);
      u32 ex = wave_incl_add(sum) - sum, tex = tin - tsum;
      shorttot = (u32)__builtin_amdgcn_readlane((int)tin, 63);
#pragma unroll
      for (u32 q = 0; q < 4u; q++) {
        W->base[4u * lane + q] = (u16)ex;
        W->obase[4u * lane + q] = pushed[q] ? (u16)0xFFFFu : (u16)tex;
        ex += c[q]; tex += t[q];
      }
    }
    wave_sync();
    {
      /* the pieces' offsets: the stack entries above this piece's own hold the value they were made for */
      u32 top = W->sp;
      top = top < DEEP_STACK ? top : DEEP_STACK;
      for (u32 i = sp - 1u + lane; i < top; i += 64u) W->st_off[i] = off + (u32)W->base[W->st_off[i]];
      wave_sync();
      if (lane == 0u) W->sp = top;
    }
    wave_sync();
    const u32 ob = shorttot ? wave_reserve(outn, shorttot) : 0u;
#ifdef DEEP_TICKS
    const u64 tb3 = wall_clock64();
#endif
    for (u32 k0 = 0; k0 < len; k0 += 256u) {
      u32 v4[4], b4[4];
      {
#pragma unroll
        for (u32 q = 0; q < 4u; q++) { const u32 k = k0 + 64u * q + lane; v4[q] = src[k < len ? k : 0u]; }
#pragma unroll
        for (u32 q = 0; q < 4u; q++) { u32 at = SA_IDX(v4[q]) + d; if (at >= n) at -= n; b4[q] = T[at]; }
      }
#pragma unroll
      for (u32 q = 0; q < 4u; q++) {
        if (k0 + 64u * q + lane >= len) continue;
        const u32 val = v4[q], by = b4[q];
        const u32 slot = atomicAdd(&W->fill[by], 1u);
        const u32 cw = W->cnt[by], c = cw & 0x7FFFFFFFu, b0 = (u32)W->base[by];
        const bool fin = c == 1u || (cw >> 31);          /* alone with its value, or of a closed sub-run: final */
        const u32 row = r0 + b0 + slot;
        if (fin || late) {                               /* see the strips' output */
          sa[row] = val | ((c > 1u && slot) ? TIE_FLAG : 0u);
          if (fin) bwt[row] = inv[SA_CODE(val)];
          if (SA_IDX(val) == 0u) M->bwt_idx = row;
        }
        if (W->obase[by] == 0xFFFFu) {
          dst[b0 + slot] = val;
        } else {
          if (c > 1u && !(cw >> 31)) {
            const u32 o = ob + (u32)W->obase[by] + slot;
            Ls.sout[o] = val; Ls.gout[o] = r0 + b0; Ls.dout[o] = d + 1u;
            if (R.build) deep_publish(R, SA_IDX(val), r0 + b0, d + 1u);
          }
          if (R.live && off + b0) R.isa[SA_IDX(val)] = ISA_ENTRY_D(r0 + b0, rank0, R.tag, d + 1u);   /* final for this launch: its rank changed */
        }
      }
    }
    if (shorttot) hmin = d + 1u < hmin ? d + 1u : hmin;
    if (__ballot(anyclosed) && lane == 0u) atomicMin(cminp, d + 1u);
    __threadfence_block();                              /* dst is read back by this wave from the next piece on */
    wave_sync();
#ifdef DEEP_TICKS
    if (lane == 0u) {                                   /* (diagnostic build) common prefix, count, scan, placement; pieces, rows */
      const u64 tb4 = wall_clock64();
      atomicAdd(&M->fticks[5], (u32)(tb1 - tb0)); atomicAdd(&M->fticks[6], (u32)(tb2 - tb1)); atomicAdd(&M->fticks[7], (u32)(tb3 - tb2));
      atomicAdd(&M->ticks[5], (u32)(tb4 - tb3)); atomicAdd(&M->ticks[6], 1u); atomicAdd(&M->ticks[7], len);
    }
#endif
  }
}

/* (deep_body's hand-over rule, below; k_bwt_long decides the same.)  Bit 31 of `handover`: the launch's long runs have had a
   launch of their own (k_bwt_long): this one passes over them. */
__device__ __forceinline__ bool deep_handed_over(const lbz_block_meta *M, u32 n, u32 round, u32 tot, u32 handover)
{
  const u32 ho0 = handover & 0xFFFFu, ho1 = (handover >> 16) & 0x7FFFu;
  return M->deep_skip || (u64)M->deep_long * 2ull > n || (ho0 && (u64)M->deep_tot[0] * 1000ull > (u64)n * ho0)
         || (round == DEEP_HANDOVER && (u64)tot * 1000ull > (u64)n * ho1);
}

/* LIVE: the launch may step by ranks (launches behind DEEP_BUILD: k_bwt_deepr); the launches up to it order by the text alone
   (k_bwt_deep) and carry none of that code -- 20 vector registers less, a wave more per SIMD */
template <bool LIVE>
__device__ __forceinline__ void deep_body(deep_lds &S, const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 segs,
                                          u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 round, u32 handover)
{
  u32 bi, seg;
  if (!seg_item(nblk, segs, &bi, &seg)) return;
  const u32 blk = lbz_round_block(first, count, bi, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (n < 2u || seg >= M->nseg) return;
  const u32 tot = M->deep_tot[round];            /* the block's tied rows that are left for this round, all segments */
  if (tot == 0u) return;
  /* a block in which k_bwt_batch left long runs tied (more than a batch of equal keys, BIG_LEVELS deep) goes to the rank
     rounds as it is */
#ifdef DEEP_DEBUG
  if (seg == 0u && threadIdx.x == 0u) printf("blk %u round %u tot %u hmin %u skip %u n %u\n", blk, round, tot, M->deep_hmin[round], M->deep_skip, n);
#endif
  /* ... and so does a block of which two fifths are still tied after the first launch (source trees, logs: repeats of
     hundreds of symbols under most rows).  Ranks double through those, but a run can only step by ranks if the rotations it
     looks up have entries, and giving every rotation one is what the rank rounds do.  Every segment decides the same.  */
  /* `handover`: the two thresholds in thousandths of the block's rows -- tied when the text rounds begin (low half; 0 = no
     such rule: what k_bwt_batch leaves tied says how many rows repeat, the second launch sees how long the repeats are) and
     tied after the first launch (high half: 400).  A block handed over here is flagged LBZ_TIES_EARLY: its rank rounds start
     behind launch DEEP_HANDOVER, on a stream of their own, beside the later text launches of the round's other blocks
     (lbz_api.hip: launch_sort).  Only launches up to DEEP_HANDOVER flag: a later one would set the flag again on a block
     whose rank rounds may have finished by then. */
  if (deep_handed_over(M, n, round, tot, handover)) {
    if (seg == 0u && threadIdx.x == 0u && round <= DEEP_HANDOVER) {
      atomicMax(&M->periodic, LBZ_TIES_EARLY); atomicMin(&M->deep_h0, M->deep_hmin[round]);
    }
    return;
  }
  const bwt_slot s = round_slot(ws, slot_bytes, ws_spill, spill_bytes, count, L, bi);
  if (round == 0u) {                              /* the bit map of rotations with a rank entry: every segment clears its share */
    u32 *map = reinterpret_cast<u32 *>(s.k0);
    const u32 words = (n + 31u) / 32u, nsg = M->nseg;
    const u32 w0 = (u32)((u64)words * seg / nsg), w1 = (u32)((u64)words * (seg + 1u) / nsg);
    for (u32 i = w0 + threadIdx.x; i < w1; i += LBZ_WG) map[i] = 0u;
  }
  const u32 m = M->seg_m[seg];
  if (m == 0u) return;
  const u64 tk0 = wall_clock64();
  const u32 tid = threadIdx.x, lane = lane_id();
  const u32 lo = M->seg_lo[seg];
  const u32 cap = bi < count ? L.cap_a : L.cap_b;
  const size_t off = lbz_elem_off(L, blk);
  const u8 *T = Tbase + off;
  u8 *bwt = Bbase + off;
  /* the list alternates between the slot's list columns and the partition's second buffers (free since k_bwt_batch) */
  u32 *colA[3] = { s.sufx + lo, s.grp + lo, s.pos + lo };
  u32 *colB[3] = { s.v1 + lo, reinterpret_cast<u32 *>(s.k1) + lo, reinterpret_cast<u32 *>(s.k1) + cap + lo };
  const u32 *sin = (round & 1u) ? colB[0] : colA[0], *gin = (round & 1u) ? colB[1] : colA[1], *din = (round & 1u) ? colB[2] : colA[2];
  u32 *sout = (round & 1u) ? colA[0] : colB[0], *gout = (round & 1u) ? colA[1] : colB[1], *dout = (round & 1u) ? colA[2] : colB[2];
  {
    u32 tot;
    const u32 f = (tid < 256u && M->inuse[tid]) ? 1u : 0u;
    const u32 ex = wg_excl_add(f, &tot, &S.sc);
    if (tot > 128u) { if (tid < 256u) S.inv[tid] = (u8)tid; }       /* bwt_setup's rule */
    else if (f) S.inv[ex] = (u8)tid;
    if (tid == 0) { S.ticket = 0; S.outn = 0u; S.h0min = 0xFFFFFFFFu; S.bad = 0; S.cmin = 0xFFFFFFFFu; }
    __syncthreads();
  }
  deep_wave *W = &S.w[wave_id()];
  if (lane < DEEP_NEAR) W->comp[64u + lane] = ~0ull;
  wave_sync();
  const deep_lists Ls = { sin, gin, din, sout, gout, dout };
  deep_ranks R;
  R.isa = s.isa; R.map = reinterpret_cast<u32 *>(s.k0);          /* the partition's key column is free since k_bwt_batch */
  R.tag = round + 1u; R.hcur = M->deep_hmin[round];
  R.build = round == DEEP_BUILD; R.live = LIVE;         /* (launch_sort: k_bwt_deepr for the launches behind DEEP_BUILD, k_bwt_deep up to it) */
#ifdef DEEP_TICKS
  u64 tkb = 0, tks = 0, tkp = 0, tko = 0, tkn = 0;          /* long runs, strip set-up, steps, output; strips */
#define DT_MARK(v) const u64 v = wall_clock64()
#define DT_ADD(acc, a, b) acc += (b) - (a)
#else
#define DT_MARK(v)
#define DT_ADD(acc, a, b)
#endif
  const u32 kmax = deep_kmax(round);
  const bool late = round + 1u == DEEP_ROUNDS || round + 1u == DEEP_HANDOVER;   /* what is tied after this launch may be for the rank rounds: they read the suffix array */
  const u32 chunk = m < 16u * DEEP_CHUNK ? 64u : DEEP_CHUNK;    /* a short list (the late launches: long repeats, a strip's steps are a chain of
                                                                   round trips) is dealt out a strip at a time */
  u32 hmin = 0xFFFFFFFFu;                               /* least depth of a run that stays tied (in the next list) */
  for (;;) {
    const u32 a = wave_claim(&S.ticket) * chunk;
    if (a >= m) break;
    const u32 e = a + chunk < m ? a + chunk : m;
    u32 p = 0;
    if (a) {                                            /* the first run that starts in [a, e): none if a long run covers them all */
      p = e;
      for (u32 w0 = a; w0 < e; w0 += 64u) {
        const u32 k = w0 + lane;
        const u32 g1 = k < m ? gin[k] : 0u, g0 = k - 1u < m ? gin[k - 1u] : 0u;
        const u64 hd = __ballot(k < e && g1 != g0);
        if (hd) { p = w0 + (u32)__ffsll((long long)hd) - 1u; break; }
      }
    }
    while (p < e) {
      DT_MARK(t0);
      const u32 k = p + lane;
      const bool have = k < m;
      u32 val = have ? sin[k] : 0u;
      const u32 g = have ? gin[k] : 0xFFFFFFFFu - lane;
      u32 d = have ? din[k] : 0u;
      const u32 dstart = d;
      const u32 gbelow = lane_from_below(g);
      const bool head0 = lane == 0u || g != gbelow;
      u32 hl = wave_incl_max(head0 ? lane : 0u);
      u32 cut = m - p <= 64u ? m - p : (u32)__builtin_amdgcn_readlane((int)hl, 63);   /* the run lane 63 sits in may go on: next strip */
      const u64 lim = __ballot(head0 && k >= e);                                       /* runs that start behind the chunk are the next claim's */
      if (lim) { const u32 f = (u32)__ffsll((long long)lim) - 1u; cut = f < cut ? f : cut; }
      if (cut == 0u) {                                                                 /* a run of 64 rows or more: not a strip's job */
        const u32 rank0 = (u32)__builtin_amdgcn_readlane((int)g, 0);
        u32 len = 64u;
        for (;;) {
          const u32 kk = p + len + lane;
          const u64 df = __ballot(kk >= m || gin[kk] != rank0);
          if (df) { len += (u32)__ffsll((long long)df) - 1u; break; }
          len += 64u;
        }
        deep_big_run(W, lane, p, len, Ls, const_cast<u32 *>(sin), s.v0 + lo, T, n, s.sa, bwt, S.inv, M, &S.outn, hmin, &S.cmin, late, R);
        p += len;
        DT_MARK(t9); DT_ADD(tkb, t0, t9);
        continue;
      }
      const bool in = lane < cut;
      /* What the list knows about a row -- its run's rank and the symbols the run shares, both as they stood when this launch
         began -- goes into the row's rank entry before the strip starts: an entry is otherwise only written when a rank
         CHANGES, so the depth of a long repeat stayed what it was at its last split, and the runs that look it up stepped
         through the repeat fifty symbols at a time (thousands of round trips in the last launches; Python sources: 5.7 ms in
         one launch for two thousand rows a block).  Tag 0 = "not of this launch": a reader takes rank and depth as they are. */
      if (in && R.live) R.isa[SA_IDX(val)] = ISA_ENTRY_D(g, g, 0u, d);
      const u32 row = g + (lane - hl);                  /* places are fixed: the run's rows are consecutive from its rank on */
      if (!in) hl = lane;
      bool tied = in, longmode = false;                  /* (the runs of a list are open: whoever listed them saw to that) */
      u64 k2 = 0;
      DT_MARK(t1); DT_ADD(tks, t0, t1);
      for (u32 step = 0; step < kmax; step++) {
        u32 at = SA_IDX(val) + d;
        if (at >= n) at -= n;
        if (LIVE) {
          /* Runs whose rows all look up rotations WITH an entry step by ranks; the others by 13 symbols of text (the
             rotation without an entry was unique by the end of launch DEEP_BUILD: the text decides soon). */
          const bool canr = tied && d < n;
          const bool has = canr && ((R.map[at >> 5] >> (at & 31u)) & 1u);
          const u64 hm = __ballot(hl == lane);                              /* first lanes of the runs as they stand */
          const u64 above = lane == 63u ? 0ull : hm >> (lane + 1u);
          const u32 he = above ? lane + 1u + (u32)__builtin_ctzll(above) : 64u;
          const u64 runmask = (he == 64u ? ~0ull : (1ull << he) - 1ull) & ~((1ull << hl) - 1ull);
          const u64 noentry = __ballot(canr && !has);
          const bool useR = canr && !(noentry & runmask);
          const bool useT = tied && !useR && d + 16u <= n;
          const u64 um = __ballot(useR || useT);
          if (!um || (round + 2u < DEEP_ROUNDS && step >= 2u && 2u * (u32)__popcll(um) < cut)) break;
          u64 s1 = 0;
          k2 = 0;
          if (useR) {
            const u64 e = R.isa[at];
            const bool fresh = ISA_TAG(e) == R.tag;
            s1 = (u64)isa_before(e, R.tag);
            /* the depth of the run the target was in when this launch began: a fresh entry's note belongs to the run it has just become
               part of, but the row was in this launch's list (depth >= hcur); any other note was written before the launch began and
               is a lower bound as it stands -- NOT to be raised to hcur: the target may have left the lists in a closed run, tied for
               good at the depth it had then */
            k2 = (u64)(fresh ? R.hcur : ISA_DEPTH(e));      /* travels with the row */
          }
          bool tsort = useT;                            /* the text runs take a slice of their next 16 bytes -- unless all agree on them */
          if (__ballot(useT)) {
            const bool mate = useT && hl != lane;
            u64 xa = 0, xb = 0;
            if (longmode && cut <= 32u && !__ballot(useR)) {
              d += deep_wide_skip(T, n, val, hl, useT, d, cut, lane);
              at = SA_IDX(val) + d;                       /* (d <= n: the skip stops where a row has come round) */
              if (at >= n) at -= n;
            }
            if (longmode && !__ballot(useT && (at + 64u > n || d + 64u > n))) {
              u64 ya[4], yb[4];
#pragma unroll
              for (u32 q = 0; q < 4u; q++) {
                ya[q] = 0; yb[q] = 0;
                if (useT) { const lbz_text16 *t16 = reinterpret_cast<const lbz_text16 *>(T + at + 16u * q); ya[q] = t16->a; yb[q] = t16->b; }
              }
              u32 same = 0;
#pragma unroll
              for (u32 q = 0; q < 4u; q++) {
                const u64 a0 = (u64)lane_from_below((u32)ya[q]) | (u64)lane_from_below((u32)(ya[q] >> 32)) << 32;
                const u64 b0 = (u64)lane_from_below((u32)yb[q]) | (u64)lane_from_below((u32)(yb[q] >> 32)) << 32;
                const bool differs = mate && (a0 != ya[q] || b0 != yb[q]);
                if (same == q && !__ballot(differs)) same = q + 1u;
              }
              if (useT) d += 16u * same;
              if (same == 4u) tsort = false;
              xa = same == 0u ? ya[0] : (same == 1u ? ya[1] : (same == 2u ? ya[2] : ya[3]));
              xb = same == 0u ? yb[0] : (same == 1u ? yb[1] : (same == 2u ? yb[2] : yb[3]));
              longmode = same != 0u;
            } else {
              if (useT) { const u64x2 x = deep_load16(T, n, SA_IDX(val), d); xa = x.x; xb = x.y; }
              const u64 a0 = (u64)lane_from_below((u32)xa) | (u64)lane_from_below((u32)(xa >> 32)) << 32;
              const u64 b0 = (u64)lane_from_below((u32)xb) | (u64)lane_from_below((u32)(xb >> 32)) << 32;
              if (!__ballot(mate && (a0 != xa || b0 != xb))) {
                if (useT) d += 16u;
                longmode = true;
                tsort = false;
              }
            }
            if (tsort) {
              const u64 hi = __builtin_bswap64(xa), lw = __builtin_bswap64(xb);
              s1 = hi >> 12;
              k2 = ((hi & 0xFFFull) << 40) | (lw >> 24);
            }
          }
          if (!__ballot(useR || tsort)) continue;       /* every run agreed on its text: nothing to order */
          deep_stage(W, lane, cut, s1, val, k2, hl, tied);
          if (__ballot(tied && tsort))                  /* useR, tsort belong to the place: runs only split */
            deep_stage(W, lane, cut, tsort ? k2 : 0ull, val, k2, hl, tied);
          deep_close(val, hl, lane, in, tied);
          /* rows that stay tied after a rank step looked up rotations of ONE run: either's note of its depth is a lower
             bound of that run's */
          W->val[lane] = (u32)k2;
          wave_sync();
          const u32 dh = W->val[hl];
          wave_sync();
          if (useR) d = d + dh < n ? d + dh : n;
          else if (tsort) d += DEEP_STEP;
          continue;
        }
        const bool can = tied && d + 16u <= n;          /* d + 16 > n: tied nearly all the way round (tiny or periodic blocks) */
        const u64 cm = __ballot(can);
        if (!cm || (round + 2u < DEEP_ROUNDS && step >= 2u && 2u * (u32)__popcll(cm) < cut)) break;   /* thinned out: the next launch packs the rest */
        u64 xa = 0, xb = 0;
        const bool mate = can && hl != lane;            /* has a row of its run in the lane below */
        if (longmode && !__ballot(can && (at + 64u > n || d + 64u > n))) {
          /* long repeats: 64 bytes at a time while every run agrees on them, 16 by 16 (no order needed for that: raw words) */
          u64 ya[4], yb[4];
#pragma unroll
          for (u32 q = 0; q < 4u; q++) {
            ya[q] = 0; yb[q] = 0;
            if (can) { const lbz_text16 *t16 = reinterpret_cast<const lbz_text16 *>(T + at + 16u * q); ya[q] = t16->a; yb[q] = t16->b; }
          }
          u32 same = 0;
#pragma unroll
          for (u32 q = 0; q < 4u; q++) {
            const u64 a0 = (u64)lane_from_below((u32)ya[q]) | (u64)lane_from_below((u32)(ya[q] >> 32)) << 32;
            const u64 b0 = (u64)lane_from_below((u32)yb[q]) | (u64)lane_from_below((u32)(yb[q] >> 32)) << 32;
            const bool differs = mate && (a0 != ya[q] || b0 != yb[q]);
            if (same == q && !__ballot(differs)) same = q + 1u;
          }
          if (can) d += 16u * same;
          if (same == 4u) continue;
          xa = same == 0u ? ya[0] : (same == 1u ? ya[1] : (same == 2u ? ya[2] : ya[3]));
          xb = same == 0u ? yb[0] : (same == 1u ? yb[1] : (same == 2u ? yb[2] : yb[3]));
          longmode = same != 0u;
        } else {
          if (can) { const u64x2 x = deep_load16(T, n, SA_IDX(val), d); xa = x.x; xb = x.y; }
          /* every run agrees on these 16 bytes: no sort, and the next step looks at 64 */
          const u64 a0 = (u64)lane_from_below((u32)xa) | (u64)lane_from_below((u32)(xa >> 32)) << 32;
          const u64 b0 = (u64)lane_from_below((u32)xb) | (u64)lane_from_below((u32)(xb >> 32)) << 32;
          if (!__ballot(mate && (a0 != xa || b0 != xb))) {
            if (can) d += 16u;
            longmode = true;
            continue;
          }
        }
        const u64 hi = __builtin_bswap64(xa), lw = __builtin_bswap64(xb);        /* big-endian: integer order = string order */
        k2 = can ? ((hi & 0xFFFull) << 40) | (lw >> 24) : 0ull;
        deep_stage(W, lane, cut, can ? hi >> 12 : 0ull, val, k2, hl, tied);
        deep_close(val, hl, lane, in, tied);
        const bool second = __ballot(tied && can) != 0ull;      /* can is a property of the place: runs only split */
        if (second) {
          deep_stage(W, lane, cut, k2, val, k2, hl, tied);
          deep_close(val, hl, lane, in, tied);
        }
        /* (no second slice: whatever is still side by side is a closed run, ordered on 52 bits -- six whole symbols; the depth a
           closed run is left at is where the rank rounds would start, and what the rank steps of the later launches add) */
        if (can) d += second ? DEEP_STEP : 6u;
      }
#ifdef DEEP_DEBUG
      if (in && (row == 3902u || row == 3903u)) printf("round %u row %u lane %u cut %u idx %u d %u tied %d hl %u g %u p %u a %u e %u m %u\n", round, row, lane, cut, SA_IDX(val), d, (int)tied, hl, g, p, a, e, m);
#endif
      DT_MARK(t2); DT_ADD(tkp, t1, t2);
      /* a row that became unique is final: its BWT byte and its suffix-array entry.  A row that is still tied is written
         again by whoever orders it; only the last launch leaves the suffix array current for the rank rounds */
      if (in && (!tied || late)) {
        s.sa[row] = val | (hl != lane ? TIE_FLAG : 0u);           /* (a row that is neither tied nor closed is its run's first lane) */
        if (!tied) bwt[row] = S.inv[SA_CODE(val)];
        if (SA_IDX(val) == 0u) M->bwt_idx = row;
      }
      {                                                 /* closed rows: not tied, not alone in their run; the least depth of such a run */
        const u64 hmf = __ballot(hl == lane);
        const bool clo = in && !tied && (hl != lane || (lane < 63u && !((hmf >> (lane + 1u)) & 1ull)));
        if (__ballot(clo)) {
          const u32 cm = wave_min(clo ? d : 0xFFFFFFFFu);
          if (lane == 0u) atomicMin(&S.cmin, cm);
        }
      }
      const u32 newrank = row - (lane - hl);
      /* (a row that stays tied and has gained depth without a change of rank -- a long repeat -- notes the depth too: the next
         launch's readers may come before its strip's refresh) */
      if (in && R.live && (newrank != g || (tied && d != dstart))) R.isa[SA_IDX(val)] = ISA_ENTRY_D(newrank, g, R.tag, d);
      const u64 tm = __ballot(tied);
      if (tm) {
        const u32 base = wave_reserve(&S.outn, (u32)__popcll(tm));
        if (tied) {
          const u32 o = base + (u32)__popcll(tm & lanes_below());
          sout[o] = val; gout[o] = newrank; dout[o] = d;
          if (R.build) deep_publish(R, SA_IDX(val), newrank, d);
          hmin = d < hmin ? d : hmin;
        }
      }
      p += cut;
      DT_MARK(t3); DT_ADD(tko, t2, t3);
#ifdef DEEP_TICKS
      tkn++;
#endif
    }
  }
#ifdef DEEP_TICKS
  if (lane == 0u && round == 0u) {
    atomicAdd(&M->fticks[0], (u32)tkb); atomicAdd(&M->fticks[1], (u32)tks); atomicAdd(&M->fticks[2], (u32)tkp);
    atomicAdd(&M->fticks[3], (u32)tko); atomicAdd(&M->fticks[4], (u32)tkn);
  }
#endif
  hmin = wave_min(hmin);
  if (lane == 0u && hmin != 0xFFFFFFFFu) atomicMin(&S.h0min, hmin);
  __syncthreads();
  if (tid == 0) {
    M->seg_m[seg] = S.outn;
    if (S.cmin != 0xFFFFFFFFu) atomicMin(&M->deep_closed, S.cmin);
    if (S.bad) M->err = 7u;
    if (S.outn) { atomicAdd(&M->deep_tot[round + 1u], S.outn); atomicMin(&M->deep_hmin[round + 1u], S.h0min); }
    if (round + 1u == DEEP_ROUNDS && S.outn) { atomicMax(&M->periodic, LBZ_TIES_LATE); atomicMin(&M->deep_h0, S.h0min); }
#ifdef DEEP_DEBUG
    if (S.outn && round + 1u == DEEP_ROUNDS) printf("blk %u seg %u left %u hmin %u\n", blk, seg, S.outn, S.h0min);
#endif
    atomicAdd(&M->sort_elems, m);
    atomicAdd(&M->fticks[8 + (round < 7u ? round : 7u)], (u32)(wall_clock64() - tk0));
  }
}

__global__ void __launch_bounds__(LBZ_WG, 6)          /* six waves a SIMD: at most 80 vector registers (five cost the text rounds 3-4 %) */
k_bwt_deep(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 segs,
           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 round, u32 handover)
{
  __shared__ deep_lds S;
  deep_body<false>(S, Tbase, Bbase, meta, L, first, count, nblk, segs, ws, slot_bytes, ws_spill, spill_bytes, slabs, round, handover);
}

__global__ void __launch_bounds__(LBZ_WG, 5)          /* (the launches behind the second: short lists, a wave less per SIMD costs them nothing measurable and leaves
                                                          room for the rank steps and the wide skip without spills) */
k_bwt_deepr(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 segs,
            u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 round, u32 handover)
{
  __shared__ deep_lds S;
  deep_body<true>(S, Tbase, Bbase, meta, L, first, count, nblk, segs, ws, slot_bytes, ws_spill, spill_bytes, slabs, round, handover);
}


/* ---- kernels 3: the rank rounds (fall-back): prefix doubling, ONE LAUNCH PER ROUND ----
 * k_bwt_fix0   every segment of a block that has ties left builds its list of tied rows (suffix, rank, row) in row order
 *              from the suffix array's tie flags -- in its own stretch [seg_lo, ..) of the slot's list columns -- and
 *              writes a rank entry for EVERY rotation of its rows.
 * k_bwt_fixr   round r, depth h = deep_h0 << r: every segment re-sorts its runs on isa[suffix + h] (doubling_round).
 *              The kernel boundary is the only synchronisation the segments of a block need: a round reads ranks of ANY
 *              rotation of the block, i.e. entries that another workgroup of the same launch may be replacing.  The
 *              invariant that makes this safe: an entry is written at most once per launch, as ONE aligned 64-bit store
 *              {rank now, rank before, tag of the launch}, and a reader takes `rank before` when the tag is the current
 *              launch's (isa_before) -- every workgroup sees the ranks as they stood when the launch began, so no run is
 *              ever ordered by a mixture of ranks from before and after a split.  A launch whose segment has nothing tied
 *              (or whose h has passed n) exits at once; the host enqueues the log2(M / 8) launches a block can need
 *              without looking.
 * k_bwt_fixend origin pointer and "exactly periodic" flag, the bytes of rows that stay tied for good.
 * `which` (round 5): the chain runs TWICE per round of blocks -- for the blocks flagged LBZ_TIES_EARLY (handed over by
 *              k_bwt_batch or by the first two text launches: a third of the blocks of a source tree, 7-12 doublings each),
 *              on a stream of its own beside the later text launches of the other blocks, and for the blocks flagged
 *              LBZ_TIES_LATE (whatever the last text launch left tied) behind both.  A chain only touches blocks with its
 *              flag, and k_bwt_fixend replaces the flag by the block's final 0 / 1.                                     */
static_assert(sizeof(u64) == 8 && alignof(u64) == 8, "a rank entry is one aligned 64-bit word");
__global__ void __launch_bounds__(LBZ_WG, 4)
k_bwt_fix0(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 segs,
           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 which)
{
  __shared__ bwt_lds S;
  u32 bi, seg;
  if (!seg_item(nblk, segs, &bi, &seg)) return;
  const u32 blk = lbz_round_block(first, count, bi, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (n < 2u || !lbz_ties_for(M->periodic, which) || seg >= M->nseg) return;
  const u32 lo = M->seg_lo[seg], hi = M->seg_lo[seg + 1u];
  if (lo >= hi) return;
  const u64 tk0 = wall_clock64();
  const bwt_slot s = seg_view(round_slot(ws, slot_bytes, ws_spill, spill_bytes, count, L, bi), lo);
  const size_t off = lbz_elem_off(L, blk);
  bwt_setup(M, &S);
  const u32 m = wg_regroup<true>(nullptr, s.sa + lo, lo, 0u, hi - lo, s, &S, Tbase + off, n, nullptr);
  if (threadIdx.x == 0) {
    M->seg_m[seg] = m;
    atomicAdd(&M->fticks[1], m);
#ifndef DEEP_TICKS
    atomicAdd(&M->fticks[6], (u32)(wall_clock64() - tk0));
#endif
  }
}

__global__ void __launch_bounds__(LBZ_WG, 4)
k_bwt_fixr(const u8 *Tbase, u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count, u32 nblk, u32 segs,
           u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 round, u32 which)
{
  __shared__ bwt_lds S;
  u32 bi, seg;
  if (!seg_item(nblk, segs, &bi, &seg)) return;
  const u32 blk = lbz_round_block(first, count, bi, slabs);
  lbz_block_meta *M = &meta[blk];
  const u32 n = M->n;
  if (n < 2u || !lbz_ties_for(M->periodic, which) || seg >= M->nseg) return;
  const u32 m = M->seg_m[seg];
  if (m == 0u) return;
  const u64 tk0 = wall_clock64();
  const u32 lo = M->seg_lo[seg];
  const bwt_slot s = seg_view(round_slot(ws, slot_bytes, ws_spill, spill_bytes, count, L, bi), lo);
  const size_t off = lbz_elem_off(L, blk);
  bwt_setup(M, &S);
  const u32 h0 = M->deep_h0 < M->deep_closed ? M->deep_h0 : M->deep_closed;      /* (closed runs are tied rows of the suffix array like the others) */
  const u64 h = (u64)h0 << round;               /* every tie left is at least h0 symbols deep, every rank at least as deep as that */
  if (h >= n) return;                           /* tied at depth >= n: tied for good (k_bwt_fixend) */
  const u32 m2 = doubling_round(Tbase + off, n, Bbase + off, s, &S, (u32)h, m, round + 1u);
  if (threadIdx.x == 0) {
    M->seg_m[seg] = m2;
    atomicMax(&M->rounds, round + 1u);
    atomicAdd(&M->sort_elems, m);
    atomicAdd(&M->fticks[0], S.bc[14]);                                       /* LDS batches of the doubling rounds */
    for (u32 i = 0; i < 4; i++) atomicAdd(&M->fticks[2 + i], S.bc[10 + i]);     /* load, run scan, per-wave sort, write-back */
#ifndef DEEP_TICKS
    atomicAdd(&M->fticks[7], (u32)(wall_clock64() - tk0));
#endif
  }
}

/* one workgroup per block of the round */
__global__ void __launch_bounds__(LBZ_WG, 4)
k_bwt_fixend(u8 *Bbase, lbz_block_meta *meta, lbz_layout L, u32 first, u32 count,
             u8 *ws, u64 slot_bytes, u8 *ws_spill, u64 spill_bytes, const u32 *slabs, u32 which)
{
  __shared__ u8 inv[256];
  __shared__ wg_scratch sc;
  const u32 tid = threadIdx.x;
  const u32 blk = lbz_round_block(first, count, blockIdx.x, slabs);
  lbz_block_meta *M = &meta[blk];
  if (M->n < 2u || !lbz_ties_for(M->periodic, which)) return;
  const bwt_slot s = round_slot(ws, slot_bytes, ws_spill, spill_bytes, count, L, blockIdx.x);
  u8 *bwt = Bbase + lbz_elem_off(L, blk);
  {
    u32 tot;
    const u32 f = (tid < 256u && M->inuse[tid]) ? 1u : 0u;
    const u32 ex = wg_excl_add(f, &tot, &sc);
    if (tot > 128u) { if (tid < 256u) inv[tid] = (u8)tid; }         /* bwt_setup's rule */
    else if (f) inv[ex] = (u8)tid;
    __syncthreads();
  }
  /* rows that are tied for good (exactly periodic block): their bytes are all equal anyway */
  u32 left = 0;
  for (u32 g = 0; g < M->nseg; g++) {
    const u32 lo = M->seg_lo[g], m = M->seg_m[g];
    left += m;
    for (u32 k = tid; k < m; k += LBZ_WG) bwt[s.pos[lo + k]] = inv[SA_CODE(s.sufx[lo + k])];
  }
  __syncthreads();
  if (tid == 0) {
    M->bwt_idx = ISA_CUR(s.isa[0]);
    M->periodic = left > 0u ? 1u : 0u;
  }
}
