/*
 * ref_probe.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin accessor layer compiled TOGETHER with the reference's own sources where
 * they lie under /root/reference/src (see oracle/Makefile, target _ref/libref.so).
 * It #include's the reference's encode.c by path so that the opaque
 * `struct encoder_state` (encode.c:62-98) becomes visible and per-stage values
 * (nblock, block bytes, bwt_idx, mtfv, selectors, code lengths ...) can be read
 * out for golden-vector generation and for pinning oracle/bz2_oracle.c.
 * No reference source text is copied into this repository.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "encode.c"   /* resolved through -I/root/reference/src */

/* ---- whole-stream driver: the compress.c call sequence (compress.c:73-118,
 * 210-250, 291-321) restated on top of the reference encoder ---- */
size_t
ref_compress_stream(const uint8_t *in, size_t len, unsigned bs100k,
                    uint8_t *out, size_t outcap)
{
  size_t mbs = (size_t)bs100k * 100000u;
  size_t o = 0;
  uint32_t combined = 0;
  if (outcap < 14) return 0;
  out[o++] = 'B'; out[o++] = 'Z'; out[o++] = 'h'; out[o++] = (uint8_t)('0' + bs100k);
  for (size_t off = 0; off < len; off += mbs) {
    const uint8_t *p = in + off;
    size_t left = len - off < mbs ? len - off : mbs;
    while (left > 0) {
      struct encoder_state *e = malloc(encoder_alloc_size(mbs));
      uint32_t crc;
      size_t before = left, size;
      encoder_init(e, mbs, CLUSTER_FACTOR);
      collect(e, p, &left);
      p += before - left;
      size = encode(e, &crc);
      if (o + (size + 3) / 4 * 4 + 10 > outcap) { free(e); return 0; }
      {
        uint32_t *buf = malloc(((size + 3) / 4) * 4);
        transmit(e, buf);
        memcpy(out + o, buf, size);
        free(buf);
      }
      o += size;
      combined = combine_crc(combined, crc);
      free(e);
    }
  }
  out[o++] = 0x17; out[o++] = 0x72; out[o++] = 0x45;
  out[o++] = 0x38; out[o++] = 0x50; out[o++] = 0x90;
  out[o++] = combined >> 24; out[o++] = combined >> 16;
  out[o++] = combined >> 8;  out[o++] = combined;
  return o;
}

/* ---- the same for -u / --sequential (compress.c:129-198 do_collect_seq): ONE encoder keeps collecting over
 * slab boundaries until its block is full, so the blocking is that of a continuous RLE1 over the input ---- */
size_t
ref_compress_seq(const uint8_t *in, size_t len, unsigned bs100k,
                 uint8_t *out, size_t outcap)
{
  size_t mbs = (size_t)bs100k * 100000u;
  size_t o = 0, pos = 0;
  uint32_t combined = 0;
  if (outcap < 14) return 0;
  out[o++] = 'B'; out[o++] = 'Z'; out[o++] = 'h'; out[o++] = (uint8_t)('0' + bs100k);
  while (pos < len) {
    struct encoder_state *e = malloc(encoder_alloc_size(mbs));
    uint32_t crc;
    size_t size;
    int full = 0;
    encoder_init(e, mbs, CLUSTER_FACTOR);
    while (!full && pos < len) {                        /* slab by slab, as the scheduler feeds it */
      size_t left = len - pos < mbs ? len - pos : mbs, before = left;
      full = collect(e, in + pos, &left);
      pos += before - left;
    }
    size = encode(e, &crc);
    if (o + (size + 3) / 4 * 4 + 10 > outcap) { free(e); return 0; }
    {
      uint32_t *buf = malloc(((size + 3) / 4) * 4);
      transmit(e, buf);
      memcpy(out + o, buf, size);
      free(buf);
    }
    o += size;
    combined = combine_crc(combined, crc);
    free(e);
  }
  out[o++] = 0x17; out[o++] = 0x72; out[o++] = 0x45;
  out[o++] = 0x38; out[o++] = 0x50; out[o++] = 0x90;
  out[o++] = combined >> 24; out[o++] = combined >> 16;
  out[o++] = combined >> 8;  out[o++] = combined;
  return o;
}

/* ---- per-stage accessors ---- */
uint32_t ref_nblock(struct encoder_state *e) { return e->nblock; }
uint32_t ref_block_crc(struct encoder_state *e) { return e->block_crc; }
int      ref_rle_state(struct encoder_state *e) { return e->rle_state; }
const uint8_t *ref_block(struct encoder_state *e)
{ return (const uint8_t *)(e->SA + e->max_block_size + GROUP_SIZE); }
const uint8_t *ref_inuse(struct encoder_state *e) { return (const uint8_t *)e->cmap; }
uint32_t ref_bwt_idx(struct encoder_state *e) { return e->bwt_idx; }
uint32_t ref_nmtf(struct encoder_state *e) { return e->nmtf; }
const uint16_t *ref_mtfv(struct encoder_state *e) { return (const uint16_t *)e->SA; }
uint32_t ref_num_selectors(struct encoder_state *e) { return e->u.s.num_selectors; }
uint32_t ref_num_trees(struct encoder_state *e) { return e->u.s.num_trees; }
unsigned ref_tree_pad(struct encoder_state *e) { return e->u.s.tree_pad; }
const uint8_t *ref_selector(struct encoder_state *e) { return e->u.s.selector; }
const uint8_t *ref_selector_mtf(struct encoder_state *e) { return e->u.s.selectorMTF; }
const uint8_t *ref_length(struct encoder_state *e, unsigned t) { return e->u.s.length[t]; }
const uint32_t *ref_code(struct encoder_state *e, unsigned t) { return e->u.s.code[t]; }
const unsigned *ref_tmap_new2old(struct encoder_state *e) { return e->u.s.tmap_new2old; }
const unsigned *ref_tmap_old2new(struct encoder_state *e) { return e->u.s.tmap_old2new; }

/* BWT of one block through the reference's divbwt (divbwt.c:1706): returns the
 * primary index, writes the n BWT bytes to out. */
int32_t
ref_bwt(const uint8_t *T, int32_t n, uint8_t *out)
{
  uint8_t *t = malloc((size_t)n + 1);
  int32_t *SA = malloc(((size_t)n + 1) * sizeof(int32_t));
  int32_t *bucket = malloc((65536 + 256) * sizeof(int32_t));
  int32_t idx, i;
  memcpy(t, T, (size_t)n);
  idx = divbwt(t, SA, bucket, n);
  for (i = 0; i < n; i++) out[i] = (uint8_t)SA[i];
  free(t); free(SA); free(bucket);
  return idx;
}

/* ---- pthreads driver on the reference encoder (cpu_mt.h): fixtures + bench.py's cpu_baseline ---- */
struct ref_mt_state { struct encoder_state *e; uint32_t *buf; };
static struct ref_mt_state *ref_mt_new(size_t mbs)
{
  struct ref_mt_state *s = malloc(sizeof *s);
  s->e = malloc(encoder_alloc_size(mbs));             /* one reusable encoder per thread */
  s->buf = malloc(((mbs + mbs / 4 + 65536) + 3) / 4 * 4);
  return s;
}
static void ref_mt_free(struct ref_mt_state *s) { free(s->e); free(s->buf); free(s); }
#ifndef CPU_MT_BLK_DEFINED
#define CPU_MT_BLK_DEFINED
typedef struct { uint32_t out_len, crc, bwt_idx, copies, slab, pad_; } cpu_mt_blk;
#endif
/* k such that T[0..n) = u^k with |u| minimal (1 if T is not a repetition): failure function */
static uint32_t block_copies(const uint8_t *T, uint32_t n)
{
  if (n < 2) return 1;
  uint32_t *f = malloc((size_t)n * sizeof *f), k = 0, r;
  f[0] = 0;
  for (uint32_t i = 1; i < n; i++) {
    while (k && T[i] != T[k]) k = f[k - 1];
    if (T[i] == T[k]) k++;
    f[i] = k;
  }
  r = n - f[n - 1];
  free(f);
  return (n % r == 0) ? n / r : 1;
}
/* ref_canon != 0: the origin pointer of an exactly periodic block (k equal rows per rotation class,
 * classes start at multiples of k) is rewritten to the smallest equal row -- the documented
 * convention of this repository; every other bit is the reference's. */
int ref_canon = 0;
static size_t ref_mt_slab(struct ref_mt_state *s, const uint8_t *in, size_t len, size_t mbs, uint8_t *out,
                          cpu_mt_blk *blk, unsigned *nblk)
{
  size_t o = 0, left = len;
  const uint8_t *p = in;
  unsigned nb = 0;
  while (left > 0) {                                  /* compress.c:93-104: leftovers are the slab's next block */
    uint32_t crc;
    size_t before = left, size;
    encoder_init(s->e, mbs, CLUSTER_FACTOR);
    collect(s->e, p, &left);
    p += before - left;
    size = encode(s->e, &crc);
    const uint32_t copies = block_copies(ref_block(s->e), s->e->nblock);
    transmit(s->e, s->buf);
    memcpy(out + o, s->buf, size);
    if (copies > 1 && ref_canon) {
      /* bits 81..104 of the block (encode.c:1185-1193: 48 magic, 32 crc, 1 randomised, 24 origin) */
      uint8_t *q = out + o + 10;
      uint32_t w = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
      uint32_t idx = (w >> 7) & 0xFFFFFFu;
      idx -= idx % copies;
      w = (w & ~(0xFFFFFFu << 7)) | (idx << 7);
      q[0] = (uint8_t)(w >> 24); q[1] = (uint8_t)(w >> 16); q[2] = (uint8_t)(w >> 8); q[3] = (uint8_t)w;
    }
    o += size;
    if (nb < 2) { blk[nb].out_len = (uint32_t)size; blk[nb].crc = crc; blk[nb].bwt_idx = s->e->bwt_idx; blk[nb].copies = copies; }
    nb++;
  }
  *nblk = nb;
  return o;
}
#define CPU_MT_NAME ref_compress_mt
#define CPU_MT_WORKER_STATE struct ref_mt_state
#define cpu_mt_state_new ref_mt_new
#define cpu_mt_state_free ref_mt_free
#define cpu_mt_slab ref_mt_slab
#include "cpu_mt.h"
