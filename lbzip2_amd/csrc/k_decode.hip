/*
 * k_decode.hip -- the inverse path: block-parallel bzip2 decoding,
 * one compressed block per workgroup, every block of a stream in flight at once.
 *
 * Follows the stages of the reference's decompressor -- scan() for block magics (src/parse.c:282),
 * retrieve() = prefix-code decoding + inverse MTF / zero-run expansion (src/decode.c:519-850),
 * decode() = the counting sort that turns the BWT string into a linked list (src/decode.c:852-942),
 * emit() = the list walk + inverse RLE1 + CRC (src/decode.c:944-1146) -- but the unit of parallelism
 * is the block, not the thread, and inside a block only what is serial BY NATURE runs on one wave: the bit cursor
 * (where a code starts depends on every code before it).  The move-to-front list is serial too, but composes: a chunk
 * of symbols run from the identity list leaves a permutation, so chunks run on all waves at once and are stitched
 * together afterwards; the pointer chase through the 3.6 MB list is done as list ranking, the counting sort is wide.
 *
 *   k_dscan   every bit position of the stream is tested for the two 48-bit magics
 *   k_dblock  one workgroup per block (k_dblock_w: 1024 threads instead of 256, for files of few blocks):
 *     dhuff_block   header, code tables, the walk from code to code -> symbols              (wave 0)
 *     dmtf_chunks   symbols -> list indices + a permutation + a decoded length per chunk     (every wave, as the symbols come)
 *     dmtf_scan     permutations composed, lengths summed                                     (wave 0)
 *     dmtf_expand   list indices -> bytes, zero runs filled -> the BWT string                 (every wave)
 *     dsort_block   cftab + stable counting sort -> tt[] (next pointer << 8 | byte)
 *     dwalk_block   the walk: RLE1'd bytes W[], decoded size and CRC (inverse RLE1 as composed state maps)
 *   k_demit   inverse RLE1 of W[] into the output at the block's offset
 */
#include "lbz_kernels.h"
#include "lbz_rand.h"

#define DEC_MAX_SEL 18002u
#define DEC_LUT_BITS 10u
#define DW_TMAX 1024u                 /* threads of a k_dblock workgroup: 256, or DW_TMAX when the file has few blocks */
#ifndef DH_HALVES
#define DH_HALVES 4                  /* a strip of the bit chain: 64 DH_HALVES bit offsets (4 against 2: +4 % on 8-bit codes, nothing on text) */
#endif
#define DM_CHUNK 1024u                /* symbols per move-to-front chunk */
#define DH_EXIT 2048u                 /* dhuff_block: a walk's stop entry with this bit leads on into the next 64 offsets */
#define DM_EOB 0xFFFFu                /* the end-of-block symbol as stored in the symbol array */

template <u32 T> struct dec_lds {
  u16 lut[LBZ_MAX_TREES][1u << DEC_LUT_BITS];   /* next 10 bits -> symbol << 5 | code length; 0: a longer code (or none) */
  int limit[LBZ_MAX_TREES][24];       /* per code length l: largest 20-bit window whose top l bits are a code of length <= l (-1: none) */
  int base[LBZ_MAX_TREES][24];        /* perm index = (window >> (20 - l)) - base */
  u16 perm[LBZ_MAX_TREES][LBZ_MAX_ALPHA + 2];
  u8 minlen[LBZ_MAX_TREES], maxlen[LBZ_MAX_TREES];
  u8 len[LBZ_MAX_TREES][LBZ_MAX_ALPHA + 2];
  u32 sel[(DEC_MAX_SEL + 7u) / 8u + 1u];        /* tree of every 50-symbol group, 4 bits each */
  u8 seq2unseq[256];
  u8 wl[T / 64u][256];             /* dmtf_expand: the list a wave's chunk starts with */
  u32 nsym, nout, err2;
  u32 cerr;                           /* what stopped the bit chain (0: the end-of-block code), kept while the symbols it did read are measured */
  u8 tbad[8];                         /* per table: 0, or the error a group that SELECTS it ends in (13: incomplete code, 6: oversubscribed; decode.c:232, :640) */
  u32 prod, fin;                      /* symbols the bit chain has handed over (a multiple of DM_CHUNK); 1 once it is done and nsym stands */
  u32 ring[260];                      /* dhuff_block: 256 dwords of the stream around the cursor (+ ring[0] again) */
  u32 ctr[2];                         /* chunk tickets of dmtf_chunks / dmtf_expand */
};

/* MSB-first bit cursor, the same in every lane of the wave (all its state is wave-uniform and lives in
 * scalar registers): the stream is fetched 256 bytes at a time -- one aligned dword per lane, a chunk ahead --
 * and the cursor takes its dwords out of the vector register with v_readlane, so the serial decoding chain
 * never waits on memory.  Dwords are aligned to the allocation, not to `in`; bytes outside [in, in + nbytes)
 * read as zero.                                                                                            */
struct ubit {
  const u32 *base;
  u64 ndw, dw;        /* dwords covering the stream; next dword to take */
  u32 tailmask;       /* valid bytes of the last dword */
  u32 cur, nxt;       /* per lane: dword (chunk * 64 + lane) of the current / the next chunk */
  u64 buf;            /* left-aligned */
  u32 live;
  u32 lead;           /* bits in front of in[0] in dword 0 */
};
__device__ __forceinline__ u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 rfl64(u64 v) { return (u64)rfl((u32)v) | (u64)rfl((u32)(v >> 32)) << 32; }
__device__ __forceinline__ u32 ub_chunk(const ubit *b, u64 chunk)          /* the raw dwords: nothing here waits for the load */
{
  const u64 i = chunk * 64u + lane_id();
  return b->base[i < b->ndw ? i : b->ndw - 1u];          /* no branch around the load: its result is first looked at a chunk later */
}
__device__ __forceinline__ u32 ub_cook(const ubit *b, u32 v, u64 chunk)    /* MSB first, once per dword instead of once per use */
{
  const u64 i = chunk * 64u + lane_id();
  if (i + 1u == b->ndw) v &= b->tailmask;
  if (i >= b->ndw) v = 0;
  return __builtin_bswap32(v);
}
__device__ __forceinline__ void ub_refill(ubit *b)
{
  while (b->live <= 32u) {
    const u32 v = (u32)__builtin_amdgcn_readlane((int)b->cur, (int)(b->dw & 63u));
    b->dw++;
    if ((b->dw & 63u) == 0u) { b->cur = ub_cook(b, b->nxt, b->dw >> 6); b->nxt = ub_chunk(b, (b->dw >> 6) + 1u); }
    b->buf |= (u64)v << (32u - b->live);
    b->live += 32u;
  }
}
__device__ __forceinline__ void ub_init(ubit *b, const u8 *in, u64 nbytes, u64 bitpos)
{
  const u32 mis = (u32)((uintptr_t)in & 3u);
  b->base = reinterpret_cast<const u32 *>(in - mis);
  b->ndw = (mis + nbytes + 3u) / 4u;
  const u32 tb = (u32)((mis + nbytes) & 3u);
  b->tailmask = tb ? (1u << (8u * tb)) - 1u : 0xFFFFFFFFu;
  b->lead = mis * 8u;
  const u64 off = bitpos + b->lead;
  b->dw = off >> 5;
  b->cur = ub_cook(b, ub_chunk(b, b->dw >> 6), b->dw >> 6);
  b->nxt = ub_chunk(b, (b->dw >> 6) + 1u);
  b->buf = 0; b->live = 0;
  ub_refill(b);
  const u32 skip = (u32)(off & 31u);
  b->buf <<= skip; b->live -= skip;
}
__device__ __forceinline__ u32 ub_get(ubit *b, u32 n)      /* 1 <= n <= 32 */
{
  if (b->live < n) ub_refill(b);
  const u32 v = (u32)(b->buf >> (64u - n));
  b->buf <<= n; b->live -= n;
  return v;
}
__device__ __forceinline__ u64 ub_bitpos(const ubit *b) { return b->dw * 32ull - b->live - b->lead; }

/* ------------------------------------------------------------------ k_dscan */
/* marks[]: bit position << 1 | kind (0 = block magic 0x314159265359, 1 = end-of-stream magic
 * 0x177245385090), in no particular order (the host sorts the handful of them). */
__global__ void __launch_bounds__(256)
k_dscan(const u8 *in, u64 nbytes, u64 *marks, u32 *nmarks, u32 cap)
{
  /* a thread takes the eight byte positions of one aligned 8-byte word (two 8-byte loads: the word and its successor)
     and tests their 64 bit offsets; grid = LBZ_DSCAN_GRID(nbytes) */
  const u32 mis = (u32)((uintptr_t)in & 7u);
  const u64 *base = reinterpret_cast<const u64 *>(in - mis);
  const u64 nwords = (mis + nbytes + 7u) >> 3;
  const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
  if (g >= nwords) return;
  u64 a = base[g], b = g + 1u < nwords ? base[g + 1u] : 0ull;
  /* bytes of these words in front of in[0] or behind in[nbytes - 1] read as zero (they are in the same aligned word as
     bytes of the stream: no fault, nothing of them is used) */
  const long long r0 = (long long)(g * 8u) - (long long)mis;           /* stream position of byte 0 of word a */
  for (u32 k = 0; k < 8u; k++) {
    const long long ra = r0 + (long long)k, rb = ra + 8;
    if (ra < 0 || (u64)ra >= nbytes) a &= ~(0xFFull << (8u * k));
    if ((u64)rb >= nbytes) b &= ~(0xFFull << (8u * k));
  }
  const u64 hi = __builtin_bswap64(a), lo = __builtin_bswap64(b);
  const u32 W[4] = { (u32)(hi >> 32), (u32)hi, (u32)(lo >> 32), (u32)lo };
  /* the first 32 bits of either magic at bit offset t of the 128 bits: one funnel shift and two compares; the rare hit
     is checked in full */
#pragma unroll
  for (u32 t = 0; t < 64u; t++) {
    const u32 top = (t & 31u) ? (W[t >> 5] << (t & 31u)) | (W[(t >> 5) + 1u] >> (32u - (t & 31u))) : W[t >> 5];
    if (top != 0x31415926u && top != 0x17724538u) continue;
    const long long p = r0 + (long long)(t >> 3);
    const u32 sh = t & 7u;
    if (p < 0 || (u64)p + (sh ? 7u : 6u) > nbytes) continue;
    const u64 w = ((t ? (hi << t) | (lo >> (64u - t)) : hi) >> 16);
    const int kind = w == 0x314159265359ull ? 0 : (w == 0x177245385090ull ? 1 : -1);
    if (kind >= 0) {
      const u32 k = atomicAdd(nmarks, 1u);
      if (k < cap) marks[k] = (((u64)p * 8ull + sh) << 1) | (u64)kind;
    }
  }
}

/* ------------------------------------------------------------------ the codes */
/* One wave per block parses the header, builds the tables (64 symbols at a time, with ballots) and walks the codes; all
 * its state is wave-uniform and the lanes are used as storage and for the wide parts.  Codes of up to 10 bits -- nearly
 * all -- resolve with one LDS lookup per bit OFFSET, 128 offsets at a time (see "symbols" below).                   */
template <u32 T> __device__ __forceinline__ void dhuff_block(const u8 *in, u64 nbytes, lbz_dblock *D, u16 *sym16, u32 cap, dec_lds<T> &S)
{
  const u32 lane = threadIdx.x;                  /* wave 0 of the workgroup */
  const u32 maxn = rfl(D->max_block < cap ? D->max_block : cap);
  ubit b;
  ub_init(&b, in, nbytes, rfl64(D->bit_start));
  u32 err = 0;
  const u32 stored_crc = ub_get(&b, 32);
  const u32 randomised = ub_get(&b, 1);
  const u32 orig_ptr = ub_get(&b, 24);
  /* used-byte map */
  const u32 big = ub_get(&b, 16);
  u32 ninuse = 0;
  for (u32 i = 0; i < 16u; i++)
    if (big & (0x8000u >> i)) {
      const u32 small = ub_get(&b, 16);
      if (lane < 16u && (small & (0x8000u >> lane))) S.seq2unseq[ninuse + (u32)__popc(small >> (16u - lane))] = (u8)(16u * i + lane);
      ninuse += (u32)__popc(small);
    }
  if (ninuse == 0u) err = 1;
  const u32 alpha = ninuse + 2u, eob = alpha - 1u;
  const u32 ngroups = ub_get(&b, 3);
  const u32 nsel = ub_get(&b, 15);
  if (!err && (ngroups < 2u || ngroups > LBZ_MAX_TREES)) err = 2;
  if (!err && nsel < 1u) err = 14;                              /* decode.c:562: no coding groups */
  /* selectors: unary move-to-front codes; the six-entry list is a word of nibbles */
  if (!err) {
    u32 order = 0x543210u, acc = 0;
    for (u32 i = 0; i < nsel; i++) {
      if (b.live < 6u) ub_refill(&b);
      const u32 v6 = (u32)(b.buf >> 58);
      const u32 j = (u32)__clz(~(v6 << 26));                       /* leading ones */
      if (j >= ngroups) { err = 3; break; }
      b.buf <<= j + 1u; b.live -= j + 1u;
      const u32 t = (order >> (4u * j)) & 15u;
      const u32 lowmask = (1u << (4u * j)) - 1u;
      order = (order & ~((lowmask << 4) | 15u)) | ((order & lowmask) << 4) | t;
      if (i < DEC_MAX_SEL) {
        acc |= t << (4u * (i & 7u));
        if ((i & 7u) == 7u || i + 1u == nsel || i + 1u == DEC_MAX_SEL) { if (lane == 0u) S.sel[i >> 3] = acc; acc = 0; }
      }
    }
  }
  /* code lengths: 5-bit start, then +1 / -1 steps (what encode.c:1231-1255 writes) */
  for (u32 t = 0; t < ngroups && !err; t++) {
    int cur = (int)ub_get(&b, 5);
    for (u32 v = 0; v < alpha && !err; v++) {
      for (;;) {
        if (cur < 1 || cur > 20) { err = 4; break; }
        if (b.live < 2u) ub_refill(&b);
        const u32 two = (u32)(b.buf >> 62);
        if (!(two & 2u)) { b.buf <<= 1; b.live -= 1u; break; }
        cur += (two & 1u) ? -1 : 1;
        b.buf <<= 2; b.live -= 2u;
      }
      if (lane == 0u) S.len[t][v] = (u8)cur;
    }
  }
  wave_sync();
  /* canonical decoding tables (codes of one length are consecutive, lengths ascend; the format's own rule),
     64 symbols at a time */
  for (u32 i = lane; i < LBZ_MAX_TREES * (1u << DEC_LUT_BITS) / 2u; i += 64u) reinterpret_cast<u32 *>(&S.lut[0][0])[i] = 0;
  wave_sync();
  for (u32 t = 0; t < ngroups && !err; t++) {
    u32 ln[5];
#pragma unroll
    for (u32 k = 0; k < 5u; k++) { const u32 v = lane + 64u * k; ln[k] = v < alpha ? S.len[t][v] : 0u; }
    u32 mn = 32, mx = 0;
#pragma unroll
    for (u32 k = 0; k < 5u; k++) if (ln[k]) { mn = ln[k] < mn ? ln[k] : mn; mx = ln[k] > mx ? ln[k] : mx; }
    mn = rfl(wave_min(mn)); mx = rfl(wave_max(mx));
    if (lane == 0u) { S.minlen[t] = (u8)mn; S.maxlen[t] = (u8)mx; }
    {
      /* the reference takes a table only if its lengths fill the code space exactly (Kraft's sum = 1, decode.c:226-235) and
         says so when a group first SELECTS the table, not when it reads it: an unused bad table is no error */
      u32 kr = 0;
#pragma unroll
      for (u32 k = 0; k < 5u; k++) if (ln[k]) kr += 1u << (20u - ln[k]);
      kr = rfl(wave_sum(kr));
      const u32 bad = kr == (1u << 20) ? 0u : (kr < (1u << 20) ? 13u : 6u);
      if (lane == 0u) S.tbad[t] = (u8)bad;
      if (bad) continue;
    }
    u32 pp = 0, vec = 0;
    for (u32 l = mn; l <= mx; l++) {
      u32 below = 0;
#pragma unroll
      for (u32 k = 0; k < 5u; k++) {
        const bool mine = ln[k] == l;
        const u64 m = __ballot(mine);
        if (mine) {
          const u32 v = lane + 64u * k;
          const u32 r = below + (u32)__popcll(m & lanes_below());
          S.perm[t][pp + r] = (u16)v;
          const u32 code = vec + r;
          if (code >> l) err = 4;                                 /* more codes than the length admits */
          else if (l <= DEC_LUT_BITS && v != eob) {               /* the end-of-block symbol takes the general step */
            const u32 e = v << 5 | l, first = code << (DEC_LUT_BITS - l);
            for (u32 j = 0; j < (1u << (DEC_LUT_BITS - l)); j++) S.lut[t][first + j] = (u16)e;
          }
        }
        below += (u32)__popcll(m);
      }
      const u32 first = vec;
      vec += below;
      if (lane == 0u) {
        S.limit[t][l] = (int)(vec << (20u - l)) - 1;              /* vec - 1 is the last code of length <= l; -1 if none */
        S.base[t][l] = (int)first - (int)pp;
      }
      pp += below;
      vec <<= 1;
    }
    err = rfl(wave_max(err));
  }
  wave_sync();
  /* symbols: the bit chain alone.  From here on the stream is read through a ring of 256 cooked dwords in LDS (four
     chunks of 64, filled a chunk ahead of the cursor; ring[256] repeats ring[0] so that a pair of dwords never wraps).
     A strip at a time: lane j looks up the 10 bits that start at bit j and at bit 64 + j behind the cursor G (one
     window read + one table read each), turns the entry's length into the offset of the code behind it, and the codes
     are then walked with one v_readlane per code (huff_walk) -- first the offsets 0..63, then 64..127; the lanes that
     were hopped on store their symbols side by side.  Nothing else is on the chain: what the symbols MEAN (move-to-
     front, zero runs) is worked out afterwards, by every wave of the workgroup (dmtf_*).                        */
  u32 nsym = 0;
  u64 G = ub_bitpos(&b) + b.lead;                                /* bit cursor in the dword stream that starts at b.base */
  if (!err) {
    u64 whi = (G >> 5) >> 6;                                     /* next chunk to put into the ring */
    for (u32 i = 0; i < 2u; i++, whi++) {
      const u32 v = ub_cook(&b, ub_chunk(&b, whi), whi);
      S.ring[((u32)whi & 3u) * 64u + lane] = v;
      if (((u32)whi & 3u) == 0u && lane == 0u) S.ring[256] = v;
    }
    u32 pf = ub_chunk(&b, whi);
    wave_sync();
    u32 groupno = 0, k = LBZ_GROUP, t = 0, pub = 0;
#ifdef DH_PROF
    u64 prof[4] = { 0, 0, 0, 0 };
#endif
    for (;;) {
      if (k == LBZ_GROUP) {                                      /* a group: LBZ_GROUP symbols of one tree */
        if (groupno >= nsel) { err = 5; break; }
        t = groupno < DEC_MAX_SEL ? (rfl(S.sel[groupno >> 3]) >> (4u * (groupno & 7u))) & 15u : 0u;
        { const u32 tb = rfl((u32)S.tbad[t]); if (tb) { err = tb; break; } }
        groupno++;
        if (nsym > maxn + 1u) { err = 8; break; }                /* every symbol but the last is at least one byte */
        k = 0;
        if ((nsym ^ pub) >= DM_CHUNK) { pub = nsym & ~(DM_CHUNK - 1u); lds_publish(&S.prod, pub); }   /* whole chunks go to the waves that wait in dmtf_chunks */
      }
      const u32 gd = (u32)(G >> 5), g5 = (u32)G & 31u;
      if ((u32)(whi * 64u - (G >> 5)) < 4u + 2u * DH_HALVES) {   /* the cursor is within a strip (+ a long code) of the ring's end */
        const u32 v = ub_cook(&b, pf, whi);
        S.ring[((u32)whi & 3u) * 64u + lane] = v;
        if (((u32)whi & 3u) == 0u && lane == 0u) S.ring[256] = v;
        whi++;
        pf = ub_chunk(&b, whi);
        wave_sync();
      }
#ifdef DH_PROF
      const u64 p0 = clock64();
#endif
      const u32 rem = LBZ_GROUP - k;
      u32 cnt = 0, used = 0, start = 0;
      bool stop = false;                                         /* the walk ended on an offset without a table entry */
#define DH_HALF(h)                                                                                            \
      u32 e##h, nx##h, nz##h, c##h = 0; u64 M##h = 0;                                                              \
      {                                                                                                        \
        const u32 pos = g5 + lane + 64u * (h), idx = (gd + (pos >> 5)) & 255u;                                 \
        const u64 w = (u64)S.ring[idx] << 32 | S.ring[idx + 1u];                                               \
        e##h = S.lut[t][(u32)((w << (pos & 31u)) >> (64u - DEC_LUT_BITS))];                                     \
        const u32 len = e##h & 31u, tgt = lane + len;                                                          \
        /* a stop: no entry here (j | 64), or the code behind starts in the next 64 offsets, at tgt - 64 (+ DH_EXIT) */ \
        nx##h = len == 0u ? lane | 64u : (tgt >= 64u ? lane | 64u | DH_EXIT | (tgt - 64u) << 7 : tgt);           \
        nz##h = (u32)__shfl((int)nx##h, (int)(nx##h & 63u));          /* two codes on; a stop stays where it is */ \
      }
      /* walk the offsets 64 h .. 64 h + 63 from `start`; the group may end there (another tree reads the bits behind it) */
#define DH_WALK(h)                                                                                            \
      {                                                                                                        \
        u32 raw;                                                                                               \
        huff_walk(nx##h, nz##h, start, raw, M##h);                                                                  \
        if (raw & DH_EXIT) { start = (raw >> 7) & 15u; used = 64u * (h + 1u) + start; }                        \
        else { M##h &= ~(1ull << (raw & 63u)); used = 64u * (h) + (raw & 63u); stop = true; }                  \
        c##h = (u32)__popcll(M##h);                                                                            \
        if (cnt + c##h >= rem) {                                                                               \
          if (cnt + c##h > rem) {                                                                              \
            const u64 q = __ballot(((M##h >> lane) & 1ull) && (u32)__popcll(M##h & lanes_below()) == rem - cnt); \
            const u32 at = (u32)__builtin_ctzll(q);                                                            \
            used = 64u * (h) + at; M##h &= (1ull << at) - 1ull; c##h = rem - cnt;                              \
          }                                                                                                    \
          cnt = rem; stop = false;                                                                             \
          goto walked;                                                                                         \
        }                                                                                                      \
        cnt += c##h;                                                                                           \
        if (stop) goto walked;                                                                                 \
      }
#if DH_HALVES == 4
      DH_HALF(0) DH_HALF(1) DH_HALF(2) DH_HALF(3)
      DH_WALK(0) DH_WALK(1) DH_WALK(2) DH_WALK(3)
    walked:
      huff_store(sym16, nsym, e0, M0);
      huff_store(sym16, nsym + c0, e1, M1);
      if (M2) huff_store(sym16, nsym + c0 + c1, e2, M2);
      if (M3) huff_store(sym16, nsym + c0 + c1 + c2, e3, M3);
#else
      DH_HALF(0) DH_HALF(1)
#ifdef DH_PROF
      const u32 keep = rfl(nx0 + nx1);
      const u64 p1 = clock64() + (keep & 0u);
#endif
      DH_WALK(0) DH_WALK(1)
    walked:
#ifdef DH_PROF
      const u64 p2 = clock64();
#endif
      huff_store(sym16, nsym, e0, M0);
      huff_store(sym16, nsym + c0, e1, M1);
#endif
#undef DH_HALF
#undef DH_WALK
      nsym += cnt; k += cnt;
      G += used;
#ifdef DH_PROF
      { const u64 p3 = clock64(); prof[0] += p1 - p0; prof[1] += p2 - p1; prof[2] += p3 - p2; prof[3]++; }
#endif
      if (!stop) continue;
      /* a long code or the end of the block: lane l tries length l -- is the 20-bit window at most the largest code of
         that length, and which symbol would it be -- and the shortest length that fits wins.  Two LDS round trips (window,
         limits and bases; then the symbols) instead of one per step of the canonical search.                        */
      const u32 i0 = (u32)(G >> 5) & 255u;
      const u64 w = (u64)S.ring[i0] << 32 | S.ring[i0 + 1u];
      const u32 code = (u32)((w << ((u32)G & 31u)) >> 44);           /* the same in every lane */
      const u32 ll = lane < 21u ? lane : 0u;
      const bool fit = lane >= S.minlen[t] && lane <= S.maxlen[t] && lane < 21u && (int)code <= S.limit[t][ll];
      const int pi = (int)(code >> (20u - ll)) - S.base[t][ll];
      const u32 sv = (fit && pi >= 0 && pi < (int)alpha) ? (u32)S.perm[t][pi] : 0xFFFFu;
      const u64 fits = __ballot(fit);
      if (fits == 0ull) { err = 6; break; }
      const u32 l = (u32)__builtin_ctzll(fits);
      const u32 sym = (u32)__builtin_amdgcn_readlane((int)sv, (int)l);
      if (sym == 0xFFFFu) { err = 6; break; }
      G += l;
      k++;
      if (lane == 0u) sym16[nsym] = sym == eob ? (u16)DM_EOB : (u16)sym;
      nsym++;
      if (sym == eob) break;
    }
#ifdef DH_PROF
    if (lane == 0u && !err && (blockIdx.x & 31u) == 0u) printf("blk %u strips %llu: fetch+lookup %llu walk %llu store+count %llu cycles per strip; symbols %u\n", blockIdx.x, prof[3], prof[0] / prof[3], prof[1] / prof[3], prof[2] / prof[3], nsym);
#endif
  }
  if (lane == 0u) {
    D->stored_crc = stored_crc; D->randomised = randomised; D->orig_ptr = orig_ptr;
    D->nblock = 0;
    D->err = err;
    D->bit_used = G - b.lead;
    /* A chain that stopped at a group (no selector left, a bad table) or on a code that no symbol has: the reference has been
       writing the runs out as it went, and if they passed the block's capacity BEFORE that point it has already said "block
       overflow" (decode.c:695, :776).  So the symbols read so far are still measured (dmtf_chunks, dmtf_scan). */
    const bool measured = err == 0u || err == 5u || err == 6u || err == 13u;
    S.cerr = err;
    S.nsym = measured ? nsym : 0u;
  }
  lds_publish(&S.fin, 1u);
}

/* ------------------------------------------------------------------ what the symbols mean */
/* The symbols of a block are RUNA/RUNB (0/1: a digit of a zero run), literals (2..: move list entry v - 1 to the
 * front and output it) and the end-of-block mark.  decode.c:519-850 does all of it in one loop, one symbol after the
 * other; here only what is serial by nature stays serial, and that part runs on every wave of the workgroup at once:
 *
 *   dmtf_chunks  a wave takes a chunk of DM_CHUNK symbols and runs the move-to-front chain over its literals
 *                starting from the IDENTITY list (the list lives in four vector registers; a front move is a
 *                wave_shr), so what it writes back for a literal is an index into the list the chunk starts with,
 *                whatever that is; what the chunk leaves behind is a permutation.  The chunk's decoded length
 *                needs no chain at all: digit k of a run is worth (d + 1) << k, and k is the number of digits
 *                right in front of it.
 *   dmtf_scan    one wave composes the permutations in order -- the real list at the start of every chunk -- and
 *                sums the lengths -- where every chunk's bytes go.
 *   dmtf_expand  every wave again, no chain: literal = list[index]; a run takes the value of the last literal in
 *                front of it (the list's front); positions from a prefix sum of the lengths.               */
__device__ __forceinline__ u32 run_digit_index(u64 R, u64 Rprev, u32 lane)
{
  u32 N = lane ? (u32)__clzll(~(R << (64u - lane))) : 0u;        /* digits right below this lane (the shifted-in zeros stop it) */
  if (N == lane) N += Rprev == ~0ull ? 64u : (u32)__clzll(~Rprev);
  return N;
}
__device__ __forceinline__ u64 run_mask_before(const u16 *sym16, u32 s0, u32 lane)
{
  const u32 v = s0 >= 64u ? sym16[s0 - 64u + lane] : 2u;
  return __ballot(v <= 1u);
}

template <u32 T> __device__ __forceinline__ void dmtf_chunks(u16 *sym16, u8 *lists, u32 *lens, dec_lds<T> &S)
{
  const u32 lane = threadIdx.x & 63u;
  for (;;) {
    const u32 c = wave_claim(&S.ctr[0]);
    const u32 s0 = c * DM_CHUNK;
    u32 s1 = s0 + DM_CHUNK;
    while (lds_observe(&S.prod) < s1) {                          /* the bit chain is still at it (wave 0 comes here when it is done) */
      if (lds_observe(&S.fin)) { const u32 nsym = rfl(S.nsym); s1 = s1 < nsym ? s1 : nsym; break; }
      wave_pause();
    }
    if (s0 >= s1) break;
    int L0 = (int)lane, L1 = (int)lane + 64, L2 = (int)lane + 128, L3 = (int)lane + 192;
    u32 clen = 0, bad = 0;
    u64 Rprev = run_mask_before(sym16, s0, lane);
    u32 vn = s0 + lane < s1 ? sym16[s0 + lane] : DM_EOB;
    for (u32 base = s0; base < s1; base += 64u) {
      const u32 v = vn;
      if (base + 64u < s1) vn = base + 64u + lane < s1 ? sym16[base + 64u + lane] : DM_EOB;
      const bool run = v <= 1u, lit = v >= 2u && v != DM_EOB;
      const u64 R = __ballot(run);
      if (run) {
        const u32 N = run_digit_index(R, Rprev, lane);
        if (N > 20u) bad = 7; else clen += (v + 1u) << N;
      } else if (lit) clen++;
      Rprev = R;
      int outv = (int)v;
      mtf_strip(L0, L1, L2, L3, v, __ballot(lit), outv, lane);
      if (lit) sym16[base + lane] = (u16)outv;
    }
    u8 *P = lists + (size_t)c * 256u;
    P[lane] = (u8)L0; P[lane + 64u] = (u8)L1; P[lane + 128u] = (u8)L2; P[lane + 192u] = (u8)L3;
    const u32 tot = wave_sum(clen);
    bad = wave_max(bad);
    if (lane == 0u) { lens[c] = tot; if (bad) atomicMax(&S.err2, bad); }
  }
}

template <u32 T> __device__ __forceinline__ void dmtf_scan(u8 *lists, u32 *lens, u32 maxn, dec_lds<T> &S)
{
  const u32 lane = threadIdx.x;                  /* wave 0 */
  const u32 nchunks = (S.nsym + DM_CHUNK - 1u) / DM_CHUNK;
  int c0 = (int)S.seq2unseq[lane], c1 = (int)S.seq2unseq[lane + 64u], c2 = (int)S.seq2unseq[lane + 128u], c3 = (int)S.seq2unseq[lane + 192u];
  u32 off = 0, err = 0;
  for (u32 c = 0; c < nchunks; c++) {
    u8 *P = lists + (size_t)c * 256u;
    const u32 p0 = P[lane], p1 = P[lane + 64u], p2 = P[lane + 128u], p3 = P[lane + 192u];
    const u32 len = rfl(lens[c]);
    P[lane] = (u8)c0; P[lane + 64u] = (u8)c1; P[lane + 128u] = (u8)c2; P[lane + 192u] = (u8)c3;
    if (lane == 0u) lens[c] = off;
    off += len;
    if (off > maxn) { err = 8; break; }
    int n0, n1, n2, n3;
#define DM_PICK(dst, p) { const int g0 = __shfl(c0, (int)((p) & 63u)), g1 = __shfl(c1, (int)((p) & 63u)), g2 = __shfl(c2, (int)((p) & 63u)), g3 = __shfl(c3, (int)((p) & 63u)); \
                          dst = (p) < 64u ? g0 : ((p) < 128u ? g1 : ((p) < 192u ? g2 : g3)); }
    DM_PICK(n0, p0) DM_PICK(n1, p1) DM_PICK(n2, p2) DM_PICK(n3, p3)
#undef DM_PICK
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  if (lane == 0u) { S.nout = off; if (err) atomicMax(&S.err2, err); }
}

template <u32 T> __device__ __forceinline__ void dmtf_expand(const u16 *sym16, const u8 *lists, const u32 *lens, u8 *tt8, dec_lds<T> &S)
{
  const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const u32 nsym = S.nsym;
  const u32 nchunks = (nsym + DM_CHUNK - 1u) / DM_CHUNK;
  u8 *wl = S.wl[w];
  for (;;) {
    const u32 c = wave_claim(&S.ctr[1]);
    if (c >= nchunks) break;
    const u32 s0 = c * DM_CHUNK, s1 = s0 + DM_CHUNK < nsym ? s0 + DM_CHUNK : nsym;
    wave_sync();
    reinterpret_cast<u32 *>(wl)[lane] = reinterpret_cast<const u32 *>(lists + (size_t)c * 256u)[lane];
    wave_sync();
    u32 pos = rfl(lens[c]);
    u32 carry = wl[0];                                          /* the list's front: what a run in front of the chunk's first literal repeats */
    u64 Rprev = run_mask_before(sym16, s0, lane);
    for (u32 base = s0; base < s1; base += 64u) {
      const u32 v = base + lane < s1 ? sym16[base + lane] : DM_EOB;
      const bool run = v <= 1u, lit = v >= 2u && v != DM_EOB;
      const u64 R = __ballot(run), Lm = __ballot(lit);
      u32 len = lit ? 1u : 0u;
      if (run) len = (v + 1u) << run_digit_index(R, Rprev, lane);     /* indices > 20 were refused by dmtf_chunks */
      Rprev = R;
      const u32 val = lit ? wl[v - 2u] : 0u;
      const u64 lower = Lm & lanes_below();
      const u32 g = (u32)__shfl((int)val, lower ? 63 - (int)__clzll(lower) : (int)lane);
      const u32 fv = lower ? g : carry;
      const u32 inc = wave_incl_add(len);
      const u32 p = pos + inc - len;
      if (lit) tt8[p] = (u8)val;
      if (run && len <= 16u) for (u32 r = 0; r < len; r++) tt8[p + r] = (u8)fv;
      u64 big = __ballot(run && len > 16u);
      while (big) {
        const u32 j = (u32)__builtin_ctzll(big);
        big &= big - 1ull;
        const u32 P = (u32)__builtin_amdgcn_readlane((int)p, (int)j), Ln = (u32)__builtin_amdgcn_readlane((int)len, (int)j);
        const u32 F = (u32)__builtin_amdgcn_readlane((int)fv, (int)j);
        for (u32 r = lane; r < Ln; r += 64u) tt8[P + r] = (u8)F;
      }
      pos += (u32)__builtin_amdgcn_readlane((int)inc, 63);
      if (Lm) carry = (u32)__builtin_amdgcn_readlane((int)val, 63 - (int)__clzll(Lm));
    }
  }
}

/* ------------------------------------------------------------------ k_dsort */
/* tt[k] = (position of the k-th byte in sorted order) << 8 | byte at position k  (decode.c:852-942).
 * Stable counting sort, 256 positions at a time: ranks inside a wave from match-any ballots, the four
 * waves of a tile in order through per-wave digit counts.                                          */
template <u32 T> struct sort_lds {
  u32 cf[256];
  u32 wcnt[T / 64u][256];
  u32 wsum[4];
};
template <u32 T> __device__ __forceinline__ void dsort_block(const lbz_dblock *D, const u8 *tt8, u32 *tt, sort_lds<T> &S)
{
  constexpr u32 NWV = T / 64u;
  u32 (&cf)[256] = S.cf;
  u32 (&wcnt)[NWV][256] = S.wcnt;
  u32 (&wsum)[4] = S.wsum;
  const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const u32 n = D->nblock;
  for (u32 i = tid; i < NWV * 256u; i += T) (&wcnt[0][0])[i] = 0;
  __syncthreads();
  for (u32 i0 = 0; i0 < n; i0 += T) {                        /* byte counts (a wave's equal bytes in one add) */
    const u32 i = i0 + tid;
    const bool ok = i < n;
    const u32 d = ok ? tt8[i] : 0u;
    u64 mask = __ballot(ok);
#pragma unroll
    for (u32 bb = 0; bb < 8u; bb++) {
      const bool bit = (d >> bb) & 1u;
      const u64 bal = __ballot(bit);
      mask &= bit ? bal : ~bal;
    }
    if (ok && (mask & lanes_below()) == 0ull) wcnt[w][d] += (u32)__popcll(mask);
  }
  __syncthreads();
  {                                                             /* the first 256 threads: one byte value each */
    u32 c = 0;
    if (tid < 256u) for (u32 k = 0; k < NWV; k++) c += wcnt[k][tid];
    u32 inc = c;
    for (u32 d = 1; d < 64u; d <<= 1) { const u32 o = (u32)__shfl_up((int)inc, d); if (lane >= d) inc += o; }
    if (tid < 256u && lane == 63u) wsum[w] = inc;
    __syncthreads();
    if (tid < 256u) {
      u32 basev = 0;
      for (u32 i = 0; i < w; i++) basev += wsum[i];
      cf[tid] = basev + inc - c;
    }
  }
  __syncthreads();
  for (u32 t0 = 0; t0 < n; t0 += T) {
    for (u32 i = tid; i < NWV * 256u; i += T) (&wcnt[0][0])[i] = 0;
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
      tt[dst] = i << 8 | tt8[dst];                     /* the byte at dst from the 0.9 MB array: no read-modify-write of the list */
    }
    __syncthreads();
    if (tid < 256u) { u32 c = 0; for (u32 k = 0; k < NWV; k++) c += wcnt[k][tid]; cf[tid] += c; }
    __syncthreads();
  }
}

/* ------------------------------------------------------------------ k_dwalk */
/* The walk (decode.c:944-1146 is one pointer chase per block: n dependent loads) done as LIST RANKING so
 * that a block has hundreds of chases in flight instead of one:
 *
 *   1. every 2^LOG-th list node (512th or 256th, see below; and the start node) is a splitter; from each splitter a lane follows the list
 *      to the next splitter and records (steps, splitter reached).  Lanes take splitters from a counter,
 *      so a long sublist does not hold the others up;
 *   2. one lane ranks the <= 1760 (3520) splitters: sublist k starts at output offset off[k].  A list that closes
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
/* splitters every 2^LOG list nodes: 512 for the 256-thread kernel; 256 for the 1024-thread one, whose blocks have the chip
   almost to themselves and wait for their LONGEST sublist (~ 2^LOG ln(n / 2^LOG) nodes, one dependent load each) */
#define DW_NONE 0xFFFFFFFFu
#define CRC_POLY 0x04C11DB7u

template <u32 T, u32 LOG> struct walk_lds {
  static constexpr u32 MAXS = (LBZ_MAX_BLOCK >> LOG) + 4u;
  u32 len[MAXS], nxt[MAXS], off[MAXS];
  u16 ord[MAXS];         /* the sublists that are on the list, longest first */
  u32 bins[24], nord;
  u32 crctab[256];
  u32 pow8[32];
  u32 p8[256], s8[256];  /* x^(8 b) and x^0 + x^8 + .. + x^(8 (b - 1)) mod P: the CRC over b copies of one byte in two multiplications */
  u32 fn[T];             /* chunk maps, 3 bits per start state */
  u32 olen[T];           /* decoded bytes of the chunk, then the exclusive prefix */
  u32 ctr, ctr2, period, total;
  u32 endstate;          /* RLE1 state behind the block's last byte: 4 = four equal bytes and no count (decode.c:1009, 1104: ERR_RUNLEN) */
  u32 xr[T / 64u];
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

template <u32 T, u32 LOG> __device__ __forceinline__ void dwalk_block(lbz_dblock *D, const u32 *tt, u8 *W, u32 *pinfo, u8 *A, u8 *X, walk_lds<T, LOG> &S)
{
  constexpr u32 DW_LOG = LOG, DW_STRIDE = 1u << LOG;
  const u32 tid = threadIdx.x;
  const u32 n = D->nblock;
  if (tid < 256u) {
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
  for (u32 i = tid; i < ns; i += T) S.off[i] = DW_NONE;
  __syncthreads();
  if (tid < 256u) S.p8[tid] = crc_shift(S.pow8, 1u, tid);
  __syncthreads();
  {                                                   /* s8[b] = p8[0] ^ .. ^ p8[b - 1] */
    const u32 pv = tid < 256u ? S.p8[tid] : 0u;
    u32 inc = pv;
    for (u32 d = 1; d < 64u; d <<= 1) { const u32 o = (u32)__shfl_up((int)inc, d); if ((tid & 63u) >= d) inc ^= o; }
    if (tid < 256u && (tid & 63u) == 63u) S.xr[tid >> 6] = inc;
    __syncthreads();
    if (tid < 256u) {
      u32 v = inc ^ pv;
      for (u32 w = 0; w < (tid >> 6); w++) v ^= S.xr[w];
      S.s8[tid] = v;
    }
  }
  __syncthreads();

  const u64 w0 = wall_clock64();
  /* 1. sublist lengths -- and the bytes met on the way: the first 2 * 2^LOG of sublist k go to A (the BWT bytes' array,
     done with) and X at k * 2^LOG, four at a time, so that the second pass (3.) copies them to their place in order
     instead of chasing the list again; where a longer sublist goes on is kept in pinfo[k] (not yet in use) */
  {
    u32 k = atomicAdd(&S.ctr, 1u);
    u32 node = k < ns0 ? k << DW_LOG : t0, cnt = 0, acc = 0;
    while (k < ns) {
      const u32 capk = k < ns0 ? 2u * DW_STRIDE : 0u;
      const u32 x = tt[node];
      if (cnt < capk) {
        acc |= (x & 255u) << (8u * (cnt & 3u));
        if ((cnt & 3u) == 3u) {
          *reinterpret_cast<u32 *>((cnt < DW_STRIDE ? A : X) + (k << DW_LOG) + (cnt & (DW_STRIDE - 1u)) - 3u) = acc;
          acc = 0;
        }
      }
      node = x >> 8;
      cnt++;
      if (cnt == capk) pinfo[k] = node;
      if ((node & (DW_STRIDE - 1u)) == 0u || node == t0 || cnt >= n) {
        if (cnt <= capk)
          for (u32 j = cnt & ~3u; j < cnt; j++) ((j < DW_STRIDE ? A : X) + (k << DW_LOG))[j & (DW_STRIDE - 1u)] = (u8)(acc >> (8u * (j & 3u)));
        S.len[k] = cnt;
        S.nxt[k] = node == t0 ? start_id : node >> DW_LOG;
        k = atomicAdd(&S.ctr, 1u);
        node = k < ns0 ? k << DW_LOG : t0;
        cnt = 0; acc = 0;
      }
    }
  }
  __syncthreads();
  const u64 w1 = wall_clock64();
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
  const u64 w2 = wall_clock64();
  /* 3. bytes: what pass 1 kept is copied to its place (a wave per sublist, in order); only sublists longer than that are
     followed again from where the copy ends -- longest first (by power of two: a counting sort over 20 bins), so that what
     a lane takes last is short; a lane collects four bytes before it stores (a store is waited for together with the next
     load) */
  if (tid < 24u) S.bins[tid] = 0;
  __syncthreads();
  for (u32 i = tid; i < ns; i += T) {
    const u32 o = S.off[i];
    if (o == DW_NONE) continue;
    const u32 capk = i < ns0 ? 2u * DW_STRIDE : 0u;
    const u32 left = o + S.len[i] > n ? n - o : S.len[i];
    if (left > capk) atomicAdd(&S.bins[31u - (u32)__clz((int)(left - capk))], 1u);
  }
  __syncthreads();
  if (tid == 0u) {
    u32 acc = 0;
    for (int b = 23; b >= 0; b--) { const u32 c = S.bins[b]; S.bins[b] = acc; acc += c; }
    S.nord = acc;
  }
  __syncthreads();
  for (u32 i = tid; i < ns; i += T) {
    const u32 o = S.off[i];
    if (o == DW_NONE) continue;
    const u32 capk = i < ns0 ? 2u * DW_STRIDE : 0u;
    const u32 left = o + S.len[i] > n ? n - o : S.len[i];
    if (left > capk) S.ord[atomicAdd(&S.bins[31u - (u32)__clz((int)(left - capk))], 1u)] = (u16)i;
  }
  __syncthreads();
  {
    const u32 nord = S.nord;
    u32 node = 0, o = 0, left = 0, acc = 0, nacc = 0;
    for (;;) {
      if (left == 0u) {
        const u32 t = atomicAdd(&S.ctr2, 1u);
        if (t >= nord) break;
        const u32 k = S.ord[t];
        const u32 capk = k < ns0 ? 2u * DW_STRIDE : 0u;
        o = S.off[k];
        left = S.len[k];
        if (o + left > n) left = n - o;
        o += capk; left -= capk;
        node = k < ns0 ? pinfo[k] : t0;
      }
      const u32 x = tt[node];
      acc |= (x & 255u) << (8u * nacc);
      nacc++;
      node = x >> 8;
      left--;
      if (((o + nacc) & 3u) == 0u || left == 0u) {
        if (nacc == 4u) *reinterpret_cast<u32 *>(W + o) = acc;
        else for (u32 j = 0; j < nacc; j++) W[o + j] = (u8)(acc >> (8u * j));
        o += nacc; acc = 0; nacc = 0;
      }
    }
  }
  for (u32 k = tid >> 6; k < ns0; k += T / 64u) {
    const u32 o = S.off[k];
    if (o == DW_NONE) continue;
    u32 m = o + S.len[k] > n ? n - o : S.len[k];
    if (m > 2u * DW_STRIDE) m = 2u * DW_STRIDE;
    for (u32 j = tid & 63u; j < m; j += 64u) W[o + j] = ((j < DW_STRIDE ? A : X) + (k << DW_LOG))[j & (DW_STRIDE - 1u)];
  }
  __syncthreads();
  const u32 period = S.period;
  if (period < n) {
    for (u32 j = period + tid; j < n; j += T) W[j] = W[j % period];
    __syncthreads();
  }

  if (D->randomised) {
    /* the format's obsolete "randomised" variant (bzip2 0.9.0 wrote it for blocks it found hard to sort; no current
       compressor does, the reference's decoder still takes it, tests/README "rand"): byte k is flipped iff k + 2 is a
       partial sum of the step table taken cyclically */
    for (u32 i = tid; i < 512u; i += T) S.nxt[i] = LBZ_RNUMS[i];
    __syncthreads();
    if (tid == 0u) { u32 acc = 0; for (u32 i = 0; i < 512u; i++) { acc += S.nxt[i]; S.nxt[i] = acc; } }
    __syncthreads();
    const u32 cyc = S.nxt[511];
    for (u32 m = tid;; m += T) {
      const u32 pos = (m >> 9) * cyc + S.nxt[m & 511u] - 2u;
      if (pos >= n) break;
      W[pos] ^= 1u;
    }
    __syncthreads();
  }

  const u64 w3 = wall_clock64();
  /* 4. chunk maps */
  const u32 npiece = (n + 15u) / 16u;
  const u32 ppc = (npiece + T - 1u) / T;        /* 16-byte pieces per chunk */
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
    for (u32 t = 0; t < T; t++) { const u32 f = S.fn[t]; S.fn[t] = c; c = (f >> (3u * c)) & 7u; }
    S.endstate = c;
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
          if (b >= 48u) crc = gf2_mulmod(crc, S.p8[b]) ^ gf2_mulmod(S.crctab[pb & 255u], S.s8[b]);
          else for (u32 r = 0; r < b; r++) crc = dec_crc_step(S.crctab, crc, pb);
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
    for (u32 t = 0; t < T; t++) { const u32 v = S.olen[t]; S.olen[t] = acc; acc += v; }
    S.total = acc;
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
    u32 cc = 0;
    for (u32 k = 0; k < T / 64u; k++) cc ^= S.xr[k];
    cc = ~cc;
    D->computed_crc = cc;
    D->out_len = total;
    /* the reference's emit() stops at a block that ends where a run's count should stand; a block it took to the end is
       checked against its CRC (expand.c:730-733: only a block whose status is OK) */
    if (S.endstate == 4u) D->err = 12;
    else if (cc != D->stored_crc) D->err = 11;
    D->wk[0] = (u32)(w1 - w0); D->wk[1] = (u32)(w2 - w1); D->wk[2] = (u32)(w3 - w2); D->wk[3] = (u32)(wall_clock64() - w3);
  }
}

/* ------------------------------------------------------------------ k_dblock */
/* The three stages of one block in one workgroup: a block whose codes are done goes on to its sort and its
 * walk while others still decode, so a pass takes the slowest block's chain, not the sum of the slowest of
 * every stage.  Wave 0 decodes (the other three wait at the barrier); sort and walk use all four.       */
template <u32 T> union dblock_lds {
  dec_lds<T> h;
  sort_lds<T> s;
  walk_lds<T, (T > 256u ? 8u : 9u)> w;
};
template <u32 T> __device__ __forceinline__ void
dblock_body(const u8 *in, u64 nbytes, lbz_dblock *blocks, u32 nblk, u8 *tt8_base, u32 *tt_base, u8 *W_base, u32 *pinfo_base, u8 *X_base, u32 cap)
{
  __shared__ dblock_lds<T> U;
  const u32 tid = threadIdx.x;
  const u32 blk = blockIdx.x;
  if (blk >= nblk) return;
  lbz_dblock *D = &blocks[blk];
  u8 *tt8 = tt8_base + (size_t)blk * cap;
  u32 *tt = tt_base + (size_t)blk * cap;
  /* scratch of the codes stage, in the block's (not yet used) list array: the symbols, then per chunk of DM_CHUNK
     symbols 256 bytes (its permutation, later the list it starts with) and a word (its decoded length, later its offset) */
  u16 *sym16 = reinterpret_cast<u16 *>(tt);
  const u32 loff = (2u * (cap + 128u) + 255u) & ~255u;
  u8 *lists = reinterpret_cast<u8 *>(tt) + loff;
  u32 *lens = reinterpret_cast<u32 *>(lists + (size_t)((cap + 128u) / DM_CHUNK + 2u) * 256u);
  const u64 k0 = wall_clock64();
  if (tid == 0u) { U.h.ctr[0] = 0; U.h.ctr[1] = 0; U.h.err2 = 0; U.h.cerr = 0; U.h.nout = 0; U.h.prod = 0; U.h.fin = 0; U.h.nsym = 0; }
  __syncthreads();
  /* wave 0 walks the codes; the others turn chunks of symbols into chunks of list indices as they come, and wave 0
     joins them when it has reached the end of the block */
  u64 ka = k0;
  if (tid < 64u) {
    const u64 c0 = clock64();
    dhuff_block(in, nbytes, D, sym16, cap, U.h);
    ka = wall_clock64();
    if (tid == 0u) D->cyc = (u32)(clock64() - c0);
  }
  dmtf_chunks(sym16, lists, lens, U.h);
  __threadfence_block();
  __syncthreads();
  u64 kb = wall_clock64();
  if (U.h.nsym) {
    if (tid < 64u && !U.h.err2) dmtf_scan(lists, lens, D->max_block < cap ? D->max_block : cap, U.h);
    __threadfence_block();
    __syncthreads();
    if (!U.h.err2 && !U.h.cerr) dmtf_expand(sym16, lists, lens, tt8, U.h);
    __syncthreads();
    if (tid == 0u) {
      u32 err = U.h.err2 ? 8u : U.h.cerr;               /* (7: a zero run of more than twenty digits, 8: more bytes than a block holds -- both "block overflow") */
      const u32 n = U.h.nout;
      if (!err && (n == 0u || D->orig_ptr >= n)) err = 9;
      D->nblock = (err && err != 9u) ? 0u : n;          /* (9 with n > 0: the origin pointer lies behind the block, ERR_BWTIDX; with n == 0: ERR_EMPTY) */
      D->err = err;
    }
  }
  __threadfence_block();
  __syncthreads();
  const u64 k1 = wall_clock64();
  if (D->err || D->nblock == 0u) {
    if (tid == 0u) { D->out_len = 0; D->tk[0] = (u32)(k1 - k0); D->tk[1] = D->tk[2] = 0; D->tk[3] = (u32)(ka - k0); D->tk[4] = (u32)(kb - ka); D->tk[5] = (u32)(k1 - kb); }
    return;
  }
  dsort_block(D, tt8, tt, U.s);
  __threadfence_block();
  __syncthreads();
  const u64 k2 = wall_clock64();
  dwalk_block(D, tt, W_base + (size_t)blk * cap, pinfo_base + (size_t)blk * (cap / 16u), tt8, X_base + (size_t)blk * cap, U.w);
  if (tid == 0u) {
    D->tk[0] = (u32)(k1 - k0); D->tk[1] = (u32)(k2 - k1); D->tk[2] = (u32)(wall_clock64() - k2);
    D->tk[3] = (u32)(ka - k0); D->tk[4] = (u32)(kb - ka); D->tk[5] = (u32)(k1 - kb);
  }
}

/* 256 threads per block when the file has more than two blocks per CU, 512 (k_dblock_m) down to one per CU, DW_TMAX when it has fewer, so that a block's sort and
 * walk -- latency-bound, a lane at a time -- have four times the lanes (the host picks, lbz_api.hip) */
__global__ void __launch_bounds__(256)
k_dblock(const u8 *in, u64 nbytes, lbz_dblock *blocks, u32 nblk, u8 *tt8_base, u32 *tt_base, u8 *W_base, u32 *pinfo_base, u8 *X_base, u32 cap)
{
  dblock_body<256u>(in, nbytes, blocks, nblk, tt8_base, tt_base, W_base, pinfo_base, X_base, cap);
}
__global__ void __launch_bounds__(512)
k_dblock_m(const u8 *in, u64 nbytes, lbz_dblock *blocks, u32 nblk, u8 *tt8_base, u32 *tt_base, u8 *W_base, u32 *pinfo_base, u8 *X_base, u32 cap)
{
  dblock_body<512u>(in, nbytes, blocks, nblk, tt8_base, tt_base, W_base, pinfo_base, X_base, cap);
}
__global__ void __launch_bounds__(DW_TMAX)
k_dblock_w(const u8 *in, u64 nbytes, lbz_dblock *blocks, u32 nblk, u8 *tt8_base, u32 *tt_base, u8 *W_base, u32 *pinfo_base, u8 *X_base, u32 cap)
{
  dblock_body<DW_TMAX>(in, nbytes, blocks, nblk, tt8_base, tt_base, W_base, pinfo_base, X_base, cap);
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
