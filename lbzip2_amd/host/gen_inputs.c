/*
 * gen_inputs.c -- deterministic stand-ins for the BASELINE.json corpora (none of enwik8/9,
 * Silesia or the kernel tarball exist on the GPU box and there is no network).  Definitions
 * from SURVEY.md section 8d / App. B4: xorshift32 (13,17,5); rand = top byte of each draw;
 * text = words of a 4096-word vocabulary picked with a squared-uniform index, separated by
 * ' ' (15/16) or '\n' (1/16).  Workload plumbing for bench.py, not part of the codec.
 */
#include <stddef.h>
#include <stdint.h>

static uint32_t xs32(uint32_t *x) { *x ^= *x << 13; *x ^= *x >> 17; *x ^= *x << 5; return *x; }

void lbzgen_rand(uint8_t *out, size_t n, uint32_t seed)
{
  uint32_t x = seed;
  for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(xs32(&x) >> 24);
}

void lbzgen_text(uint8_t *out, size_t n, uint32_t seed)
{
  static __thread char words[4096][10];
  static __thread uint8_t wlen[4096];
  uint32_t x = seed;
  size_t o = 0;
  for (int w = 0; w < 4096; w++) {
    wlen[w] = (uint8_t)(2 + xs32(&x) % 8);
    for (int k = 0; k < wlen[w]; k++) words[w][k] = (char)('a' + xs32(&x) % 26);
  }
  while (o < n) {
    uint32_t k = xs32(&x) % 4096;
    k = (k * k) >> 12;
    for (int i = 0; i < wlen[k] && o < n; i++) out[o++] = (uint8_t)words[k][i];
    if (o < n) out[o++] = (xs32(&x) % 16 == 0) ? '\n' : ' ';
  }
}
