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
 *   -f IN -o OUT   file splitter / muxer at line rate (process.c:260-307 source thread,
 *             :351-417 sink thread, compress.c:238-250 reorder) -- lbzamd_io.c: -R reader threads fill a ring of
 *             page-locked chunk buffers (-c slabs per chunk; pread at chunk offsets on a regular file), -p pipeline
 *             threads -- each with its own device context, so the H2D, kernels and D2H of consecutive chunks
 *             overlap -- compress chunks as body-only slab ranges (lbzamd_compress_host_body), and -W writer threads
 *             put every finished chunk whose offset is known in place (pwrite; a pipe: one writer, in order): header,
 *             bodies, trailer with the CRC folded from the 12-byte partials (lbzamd_fold_parts).  The file is never
 *             resident as a whole; the stream is the same single stream the batch call and reference lbzip2 produce.
 *
 *   -g N      (with -f/-o) the pipelines' contexts live on N devices, pipeline i on device i mod N (0 = every
 *             device; -p then counts pipelines PER device): the chunks go to the GPUs by direct H2D from the pinned
 *             ring -- the C-side form of the reference's N workers behind one splitter/muxer (process.c:515-548,
 *             compress.c:73-118, :238-250).
 *
 *   lbzamd_compress [-1..-9] [-w N] < input > output.bz2
 *   lbzamd_compress [-1..-9] -f input -o output.bz2 [-c slabs] [-p pipelines] [-g devices] [-t]
 *
 * A test driver, not the command: lbzip2's option surface is lbzamd.c.  This one exists so that the
 * drop-in boundary is exercised from C the way the reference would.
 */
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/lbzip2_amd.h"
#include "lbzamd_io.h"

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

/* ------------------------------------------------------------------ -f/-o: streaming splitter + muxer (lbzamd_io.c) */
static int stream_files(const char *in_path, const char *out_path, unsigned level, unsigned chunk_slabs, unsigned npipes, int timing,
                        unsigned ndev, unsigned readers, unsigned writers)
{
  const int fd_in = strcmp(in_path, "-") ? open(in_path, O_RDONLY) : 0;
  const int fd_out = strcmp(out_path, "-") ? open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644) : 1;
  if (fd_in < 0 || fd_out < 0) { perror("open"); return 1; }
  struct lbzamd_io_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.level = level; cfg.chunk_slabs = chunk_slabs; cfg.pipes = npipes; cfg.ndev = ndev; cfg.readers = readers; cfg.writers = writers;
  cfg.report = timing;
  int sys = 0;
  char msg[256];
  const int rc = lbzamd_io_compress(fd_in, fd_out, &cfg, NULL, &sys, msg, sizeof msg);
  if (rc) fprintf(stderr, "lbzamd: %s%s%s\n", msg, sys ? ": " : "", sys ? strerror(sys) : "");
  if (fd_in > 0) close(fd_in);
  if (fd_out > 1 && close(fd_out)) { perror("close"); return 1; }
  return rc ? 1 : 0;
}

int main(int argc, char **argv)
{
  unsigned level = 9, nworkers = 0, timing = 0, repeat = 1, chunk_slabs = 0, npipes = 2, decompress = 0, sequential = 0, ndev = 0, readers = 0, writers = 0;
  int want_dev = -1;
  const char *in_path = NULL, *out_path = NULL;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-f") && i + 1 < argc) { in_path = argv[++i]; continue; }
    if (!strcmp(argv[i], "-o") && i + 1 < argc) { out_path = argv[++i]; continue; }
    if (!strcmp(argv[i], "-c") && i + 1 < argc) { chunk_slabs = (unsigned)atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "-p") && i + 1 < argc) { npipes = (unsigned)atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "-g") && i + 1 < argc) { want_dev = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "-R") && i + 1 < argc) { readers = (unsigned)atoi(argv[++i]); continue; }   /* reader / writer threads of -f/-o */
    if (!strcmp(argv[i], "-W") && i + 1 < argc) { writers = (unsigned)atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "-t")) { timing = 1; continue; }            /* phase times on stderr */
    if (!strcmp(argv[i], "-d")) { decompress = 1; continue; }        /* the inverse path: .bz2 -> bytes */
    if (!strcmp(argv[i], "-u")) { sequential = 1; continue; }        /* the reference's -u: blocks cut where they are full (batch mode) */
    if (!strcmp(argv[i], "-r") && i + 1 < argc) { repeat = (unsigned)atoi(argv[++i]); continue; }   /* run the codec phase N times */
    if (argv[i][0] == '-' && argv[i][1] >= '1' && argv[i][1] <= '9' && !argv[i][2]) level = argv[i][1] - '0';
    else if (!strcmp(argv[i], "-w") && i + 1 < argc) nworkers = (unsigned)atoi(argv[++i]);
    else { fprintf(stderr, "usage: %s [-1..-9] [-w N] [-t] [-r N] < in > out.bz2 | -f IN -o OUT [-c slabs] [-p pipelines] [-g devices] | -d [-f IN] [-o OUT]\n", argv[0]); return 2; }
  }
  if (decompress) {
    /* whole file in, every block decoded at once (lbzamd_decompress_alloc), whole file out */
    FILE *fi = in_path && strcmp(in_path, "-") ? fopen(in_path, "rb") : stdin;
    FILE *fo = out_path && strcmp(out_path, "-") ? fopen(out_path, "wb") : stdout;
    if (!fi || !fo) { perror("lbzamd_compress -d"); return 1; }
    size_t zlen;
    unsigned char *z = read_all(fi, &zlen);
    lbzamd_dctx *d;
    const double t0 = now_s();
    unsigned maxb = (unsigned)(zlen / 20000 + 8);
    if (lbzamd_dcreate(&d, -1, maxb > 2400 ? 2400 : maxb)) { fprintf(stderr, "lbzamd: %s\n", lbzamd_last_error()); return 1; }
    unsigned char *out = NULL;
    size_t n = 0;
    const double t1 = now_s();
    if (lbzamd_decompress_alloc(d, z, zlen, &out, &n)) { fprintf(stderr, "lbzamd: %s\n", lbzamd_last_error()); return 1; }   /* one pass: the output buffer grows */
    const double t2 = now_s();
    if (timing) {
      lbzamd_dstats ds;
      lbzamd_dget_stats(d, &ds);
      fprintf(stderr, "decode: %zu B -> %zu B, %u blocks in %u stream(s); context %.3f s, decode %.3f s = %.0f MB/s (device: scan %.1f blocks %.1f emit %.1f ms)\n",
              zlen, n, ds.nblocks, ds.nstreams, t1 - t0, t2 - t1, (double)n / (t2 - t1) / 1e6, ds.ms_scan, ds.ms_blocks, ds.ms_emit);
    }
    if (fwrite(out, 1, n, fo) != n || fflush(fo)) { perror("lbzamd_compress -d: write"); return 1; }
    if (fo != stdout && fclose(fo)) { perror("lbzamd_compress -d: close"); return 1; }
    lbzamd_ddestroy(d);
    lbzamd_free(out); free(z);
    return 0;
  }
  if (in_path || out_path) {
    if (npipes < 1) npipes = 1;
    if (want_dev >= 0) {
      const int have = lbzamd_device_count();
      if (have < 1) { fprintf(stderr, "lbzamd: no HIP device\n"); return 1; }
      ndev = (unsigned)(want_dev == 0 || want_dev > have ? have : want_dev);
    }
    return stream_files(in_path ? in_path : "-", out_path ? out_path : "-", level, chunk_slabs, npipes, (int)timing, ndev, readers, writers);
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
    if (sequential && lbzamd_set_sequential(ctx, 1)) { fprintf(stderr, "lbzamd: %s\n", lbzamd_last_error()); return 1; }
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
