/*
 * lbz_common.h -- format constants and the HBM data layout shared by the host runtime
 * and the gfx950 kernels.
 *
 * Scope: the per-block compressor of lbzip2's src/encode.c +
 * src/divbwt.c.  One bzip2 block is owned by ONE workgroup in every kernel; a slab
 * (bs100k*100000 input bytes, process.c:631) yields a primary block and, when RLE1
 * expanded it, one small spill block (compress.c:98-104).
 */
#ifndef LBZ_COMMON_H
#define LBZ_COMMON_H

#include <stddef.h>
#include <stdint.h>

/* bzip2 format constants (values of common.h:42-52, they ARE the format) */
#define LBZ_MAX_ALPHA   258u
#define LBZ_MAX_TREES   6u
#define LBZ_GROUP       50u
#define LBZ_MAX_CODELEN 20u
#define LBZ_MAX_BLOCK   900000u
#define LBZ_MAX_SEL     18002u       /* ceil(900001/50) + pad selector */
#define LBZ_RUN_CAP     259u         /* 4 + 255, encode.c:105 */
#define LBZ_CLUSTER     8u           /* EM iterations, encode.h:22 */

/* workgroup geometry: 16 waves of 64 lanes on one CU */
#ifndef LBZ_WG
#define LBZ_WG 1024
#endif
#define LBZ_NW (LBZ_WG / 64)
#ifndef LBZ_COLLECT_WG
#define LBZ_COLLECT_WG 512  /* k_collect's own geometry: scans and barriers, two workgroups per CU wait less on each other (-13 %) */
#endif
#ifndef LBZ_MTF_WG
#define LBZ_MTF_WG 1024     /* k_mtf: a wave per slice, sixteen slices per block (512: 7.5 -> 8.0 ms alone, -0.5 % on three streams; 256: -1 %) */
#endif
#ifndef LBZ_ENCODE_WG
#define LBZ_ENCODE_WG 512   /* k_encode's own geometry: 74 KB of LDS and eight waves, so that two workgroups share a CU -- one's serial stretches
                               (a Huffman merge is one lane for 80 us) run beside the other's parallel ones -- and a workgroup finds room
                               beside the sorters' (one 141 KB, sixteen-wave workgroup needed a CU to itself) */
#endif
#define LBZ_HEAD_SLABS 96u  /* host-buffer calls: slabs of the short first round (the device starts after 1.5 ms of PCIe traffic) */
#define LBZ_FINISH_WG 256   /* k_offsets / k_gather: small workgroups -- a sixteen-wave workgroup that needs a whole CU's wave slots
                               at once waits tens of milliseconds behind the sorters' small workgroups (profiles/r03_h_host_timeline.txt) */
#ifndef LBZ_BWT_WG
#define LBZ_BWT_WG 256      /* the BWT kernels' own geometry: 4 waves, 1024-row batches (37 KB of LDS), four workgroups per CU.
                               The sorting kernels spend 70-80 % of their wave cycles waiting (LDS round trips, workgroup
                               barriers, gathers: profiles/r03_b_pmc_summary.json); more, smaller barrier domains per CU wait
                               less on each other: 1024 threads -> 512: +10 % (round 2), 512 -> 256: +7 % (round 3, with a
                               32-bit partition so that fewer groups outgrow the smaller batch) */
#endif

#ifndef LBZ_BWT_SEGS
#define LBZ_BWT_SEGS 32u    /* segments of a block's sorted rows = workgroups per block in k_bwt_batch / k_bwt_deep / k_bwt_fix*: fewer blocks at a time
                               per XCD, so more of their text in its L2 for the text rounds (16: wiki -3 %, Python sources -8 %) */
#endif
#ifndef LBZ_DEEP_ROUNDS
#define LBZ_DEEP_ROUNDS 8u  /* launches of k_bwt_deep (the text rounds) per round of blocks */
#endif
#define LBZ_BWT_MAXSEGS 32u /* ... and in rounds of fewer blocks than CUs, where a block's chain of launches is what the caller waits for */

/* Per-block record in HBM.  Blocks are numbered 2*slab (primary) and 2*slab+1 (spill). */
typedef struct lbz_block_meta {
  uint32_t n;          /* RLE1'd length (nblock); 0 = block absent */
  uint32_t crc;        /* running CRC of the consumed raw bytes, un-inverted (encode.c:542) */
  uint32_t consumed;   /* raw bytes consumed */
  uint32_t bwt_idx;    /* row of rotation 0 (smallest equal row if exactly periodic) */
  uint32_t periodic;   /* 1 if rows were still tied at depth >= n (T = u^k) */
  uint32_t nmtf;       /* MTF/ZRLE symbols incl. EOB */
  uint32_t alpha;      /* alphabet size = EOB + 1 */
  uint32_t num_trees;
  uint32_t num_sel;    /* incl. pad selector */
  uint32_t out_len;    /* bytes; every block is byte aligned (encode.c:514-525) */
  uint32_t err;        /* non-zero: internal capacity problem */
  uint32_t rounds;     /* prefix-doubling rounds run (diagnostic) */
  uint32_t sort_elems; /* sum of elements passed through the radix sorter (diagnostic) */
  uint32_t deep_rows;  /* (diagnostic) tied rows k_bwt_batch handed to the text rounds (k_bwt_deep), summed over the segments */
  /* The sorted rows of a block are cut into up to LBZ_BWT_SEGS segments at boundaries of the partition's groups; from
     k_bwt_batch on every (block, segment) is a workgroup of its own (k_bwt.hip).  Rows [seg_lo[s], seg_lo[s+1]).    */
  uint32_t nseg;
  uint32_t msd_bits;    /* key bits the block was partitioned on in HBM (k_bwt_part decides: 32, or 16 for incompressible data) */
  uint32_t deep_h0;     /* depth every tie that is left for the rank rounds (k_bwt_fix*) is known to reach: they double from here.
                           0xFFFFFFFF while nothing is left (k_bwt_batch and the last k_bwt_deep launch lower it) */
  uint32_t deep_long;   /* tied rows of the block in runs of more than 63 (k_bwt_batch counts them): a block that is mostly such runs skips the text rounds */
  uint32_t deep_closed; /* least depth at which a run of tied rows was left as it is because its order cannot show in the output: every row
                           has the same byte in front of it and none is rotation 0 (k_bwt.hip, "closed runs").  The suffix array keeps
                           such rows tied, so the rank rounds -- should the block still need them -- double from min(deep_h0, deep_closed) */
  uint32_t deep_skip;   /* k_bwt_batch left long runs tied (BIG_ROUNDS refinements did not split them): the block goes to the rank rounds as it is */
  uint32_t deep_tot[LBZ_DEEP_ROUNDS + 1];   /* tied rows of the block that enter text round r (its segments' lists together); [LBZ_DEEP_ROUNDS]: left over */
  uint32_t deep_hmin[LBZ_DEEP_ROUNDS + 1];  /* ... and the least depth any of them is known to share */
  uint32_t seg_lo[LBZ_BWT_MAXSEGS + 1];
  uint32_t seg_m[LBZ_BWT_MAXSEGS];         /* length of the segment's list of tied rows (k_bwt_batch -> k_bwt_deep rounds; k_bwt_fix* build their own) */
  uint32_t ticks[8];   /* wall_clock64 ticks of k_bwt_part / k_bwt_batch phases (diagnostic) */
  uint32_t fticks[16];  /* wall_clock64 ticks of k_bwt_fix phases (diagnostic) */
  uint8_t  inuse[256]; /* used-byte map (encode.c:63) */
} lbz_block_meta;

/* Arithmetic layout of the per-block arrays.  cap_a / cap_b are element capacities of a
 * primary / spill block, rounded up so every block starts 256-byte aligned.           */
typedef struct lbz_layout {
  uint32_t M;          /* bs100k * 100000 */
  uint32_t cap_a;      /* >= M + 64 */
  uint32_t cap_b;      /* >= M/4 + 64: a spill holds at most ~0.25 M bytes */
  uint32_t out_a;      /* bytes of compressed output reserved per primary block */
  uint32_t out_b;
} lbz_layout;

#if defined(__cplusplus)
static inline
#if defined(__HIPCC__) || defined(LBZ_EMULATED)
__host__ __device__
#endif
size_t lbz_elem_off(const lbz_layout L, uint32_t blk)
{
  return (size_t)(blk >> 1) * ((size_t)L.cap_a + L.cap_b) + ((blk & 1u) ? L.cap_a : 0u);
}
static inline
#if defined(__HIPCC__) || defined(LBZ_EMULATED)
__host__ __device__
#endif
size_t lbz_out_off(const lbz_layout L, uint32_t blk)
{
  return (size_t)(blk >> 1) * ((size_t)L.out_a + L.out_b) + ((blk & 1u) ? L.out_a : 0u);
}
#endif

/* BWT workspace of one resident workgroup ("slot"), elements of capacity cap_a:
 *   k0,k1 : u64 sort keys (ping-pong)      v0,v1 : u32 sort values (ping-pong)
 *   sufx,grp,pos : u32 active-list columns  sa : u32 suffix array  isa : u64 rank entries (k_bwt.hip, ISA_ENTRY)
 *   gb : 32769 u32 bucket starts                                                      */
#define LBZ_BWT_SLOT_BYTES(cap) ((size_t)(cap) * (8u * 2u + 4u * 2u + 4u * 4u + 8u) + 33024u * 4u)

#endif
