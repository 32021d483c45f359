/*
 * lbzamd_compress.c -- minimal C host driver over the C ABI (include/lbzip2_amd.h).
 *
 * It reproduces what lbzip2's pipeline does around the block codec -- slab split
 * (src/process.c:631), work units collect -> encode -> transmit (src/compress.c:73-118,
 * 210-228), in-order mux + CRC fold (compress.c:238-250), header/trailer (compress.c:291-321)
 * -- in two ways:
 *
 *   default   one call of the batch interface (everything on the device);
 *   -w N      the reference's own work-unit interface driven by N pthreads, exactly the calls
 *             compress.c makes, each on its own encoder_state (compress.c:81-115 runs them
 *             outside the scheduler lock, concurrently).
 *
 *   lbzamd_compress [-1..-9] [-w N] < input > output.bz2
 *
 * Not a CLI clone of lbzip2; it exists so the drop-in boundary is exercised
 * from C the way the reference would.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/lbzip2_amd.h"

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static unsigned char *read_all(FILE *f, size_t *len)
{
  size_t cap = 1u << 20, n = 0;
  unsigned char *b = malloc(cap);
  for (;;) {
    size_t r = fread(b + n, 1, cap - n, f);
    n += r;
    if (r == 0) break;
    if (n == cap) { cap *= 2; b = realloc(b, cap); }
  }
  *len = n;
  return b;
}

struct unit { const unsigned char *p; size_t len; void *out; size_t size; uint32_t crc; struct unit *more; };
struct job { struct unit *units; size_t nunits, next; unsigned long mbs; pthread_mutex_t mu; };

/* one work unit = one slab: (collect, encode, transmit) until the slab is consumed */
static void do_slab(struct unit *u, unsigned long mbs)
{
  const unsigned char *p = u->p;
  size_t left = u->len;
  struct unit *cur = u;
  for (struct unit *m = u->more; m;) { struct unit *nx = m->more; free(m->out); free(m); m = nx; }   /* a repeated pass (-r) */
  u->more = NULL;
  free(u->out);
  u->out = NULL;
  while (left > 0) {
    struct encoder_state *e = malloc(encoder_alloc_size(mbs));
    size_t before = left;
    encoder_init(e, mbs, CLUSTER_FACTOR);
    collect(e, p, &left);
    p += before - left;
    cur->size = encode(e, &cur->crc);
    cur->out = malloc((cur->size + 3) / 4 * 4);
    transmit(e, cur->out);
    free(e);
    if (left > 0) { cur->more = calloc(1, sizeof *cur); cur = cur->more; }
  }
}

static void *worker(void *arg)
{
  struct job *j = arg;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    size_t i = j->next++;
    pthread_mutex_unlock(&j->mu);
    if (i >= j->nunits) return NULL;
    do_slab(&j->units[i], j->mbs);
  }
}

int main(int argc, char **argv)
{
  unsigned level = 9, nworkers = 0, timing = 0, repeat = 1;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-t")) { timing = 1; continue; }            /* phase times on stderr */
    if (!strcmp(argv[i], "-r") && i + 1 < argc) { repeat = (unsigned)atoi(argv[++i]); continue; }   /* run the codec phase N times */
    if (argv[i][0] == '-' && argv[i][1] >= '1' && argv[i][1] <= '9' && !argv[i][2]) level = argv[i][1] - '0';
    else if (!strcmp(argv[i], "-w") && i + 1 < argc) nworkers = (unsigned)atoi(argv[++i]);
    else { fprintf(stderr, "usage: %s [-1..-9] [-w N] [-t] [-r N] < in > out.bz2\n", argv[0]); return 2; }
  }
  size_t len;
  unsigned char *in = read_all(stdin, &len);
  const unsigned long mbs = level * 100000ul;

  if (nworkers == 0) {
    lbzamd_ctx *ctx;
    size_t nslabs = (len + mbs - 1) / mbs, cap = lbzamd_bound(len), n = 0;
    unsigned char *out = malloc(cap);
    double t0 = now_s();
    if (lbzamd_create(&ctx, -1, level, nslabs ? (unsigned)(nslabs > 1200 ? 1200 : nslabs) : 1, 0)) {
      fprintf(stderr, "lbzamd: %s\n", lbzamd_last_error());
      return 1;
    }
    double t1 = now_s();
    for (unsigned r = 0; r < repeat; r++) {
      if (lbzamd_compress_host(ctx, in, len, out, cap, &n)) { fprintf(stderr, "lbzamd: %s\n", lbzamd_last_error()); return 1; }
      const double t2 = now_s();
      if (timing) fprintf(stderr, "batch interface: context %.3f s, pass %u: %.3f s = %.0f MB/s\n", t1 - t0, r, t2 - t1, (double)len / (t2 - t1) / 1e6);
      t1 = t2;
    }
    fwrite(out, 1, n, stdout);
    lbzamd_destroy(ctx);
    free(out);
  } else {
    struct job j = { 0 };
    j.nunits = (len + mbs - 1) / mbs;
    j.units = calloc(j.nunits ? j.nunits : 1, sizeof *j.units);
    j.mbs = mbs;
    pthread_mutex_init(&j.mu, NULL);
    for (size_t i = 0; i < j.nunits; i++) {
      j.units[i].p = in + i * mbs;
      j.units[i].len = (i + 1) * mbs <= len ? mbs : len - i * mbs;
    }
    pthread_t *th = malloc(nworkers * sizeof *th);
    for (unsigned r = 0; r < repeat; r++) {
      const double t0 = now_s();
      j.next = 0;
      for (unsigned t = 0; t < nworkers; t++) pthread_create(&th[t], NULL, worker, &j);
      for (unsigned t = 0; t < nworkers; t++) pthread_join(th[t], NULL);
      const double t1 = now_s();
      if (timing) fprintf(stderr, "work-unit interface, %u threads, pass %u: %.3f s = %.0f MB/s\n", nworkers, r, t1 - t0, (double)len / (t1 - t0) / 1e6);
    }
    /* in-order mux, compress.c:238-250 */
    unsigned char hdr[HEADER_SIZE] = { 'B', 'Z', 'h', (unsigned char)('0' + level) };
    uint32_t cc = 0;
    fwrite(hdr, 1, HEADER_SIZE, stdout);
    for (size_t i = 0; i < j.nunits; i++)
      for (struct unit *u = &j.units[i]; u; u = u->more) {
        fwrite(u->out, 1, u->size, stdout);
        cc = combine_crc(cc, u->crc);
      }
    unsigned char tr[TRAILER_SIZE] = { 0x17, 0x72, 0x45, 0x38, 0x50, 0x90,
                                       (unsigned char)(cc >> 24), (unsigned char)(cc >> 16),
                                       (unsigned char)(cc >> 8), (unsigned char)cc };
    fwrite(tr, 1, TRAILER_SIZE, stdout);
  }
  return 0;
}
