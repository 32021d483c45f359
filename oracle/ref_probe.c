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
