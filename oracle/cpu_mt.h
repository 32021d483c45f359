/*
 * cpu_mt.h -- TEST INFRASTRUCTURE (oracle/): a pthreads driver that compresses a buffer slab by
 * slab on N host threads, the way the reference's compress.c/process.c do (slab split
 * process.c:631, one encoder per worker compress.c:73-118, in-order mux + CRC fold
 * compress.c:238-250, header/trailer :291-321).  #include'd by ref_probe.c (on the reference's
 * own encode.c -> "reference") and by bz2_oracle.c (on the restatement -> "port"); the including
 * file defines
 *
 *     CPU_MT_NAME                      exported function name
 *     CPU_MT_WORKER_STATE              per-thread state type (one reusable encoder per thread)
 *     cpu_mt_state_new(mbs) / cpu_mt_state_free(st)
 *     cpu_mt_slab(st, in, len, mbs, out, blk)  -> bytes written; fills blk[0..nb) and returns nb via *nblk
 *
 * Used for (a) bench.py's cpu_baseline leg, (b) fixture generation.  Never linked into the product.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef CPU_MT_BLK_DEFINED
#define CPU_MT_BLK_DEFINED
typedef struct {
  uint32_t out_len, crc;        /* crc: un-inverted running CRC (encode.c:542) */
  uint32_t bwt_idx, copies;     /* origin pointer as emitted; copies = k if the RLE1'd block is u^k (1 = not periodic) */
  uint32_t slab, pad_;          /* slab the block came from */
} cpu_mt_blk;
#endif

struct cpu_mt_job {
  const uint8_t *in;
  size_t len, mbs, nslabs;
  size_t next;                    /* next slab to hand out (under mu) */
  pthread_mutex_t mu;
  uint8_t **sout;                 /* per slab: its packed blocks */
  size_t *slen;
  cpu_mt_blk (*sblk)[2];          /* a slab yields at most two blocks (compress.c:98-104) */
  uint8_t *snb;
};

static void *cpu_mt_worker(void *arg)
{
  struct cpu_mt_job *j = arg;
  CPU_MT_WORKER_STATE *st = cpu_mt_state_new(j->mbs);
  for (;;) {
    pthread_mutex_lock(&j->mu);
    const size_t s = j->next++;
    pthread_mutex_unlock(&j->mu);
    if (s >= j->nslabs) break;
    const size_t off = s * j->mbs;
    const size_t n = j->len - off < j->mbs ? j->len - off : j->mbs;
    uint8_t *o = malloc(n + n / 4 + 65536);
    unsigned nb = 0;
    j->slen[s] = cpu_mt_slab(st, j->in + off, n, j->mbs, o, j->sblk[s], &nb);
    j->snb[s] = (uint8_t)nb;
    j->sout[s] = o;
  }
  cpu_mt_state_free(st);
  return NULL;
}

/* Returns the stream length (0: out too small).  blocks (may be NULL) receives one record per
 * block in stream order, *nblocks their number; *seconds the wall time of compression + mux. */
size_t CPU_MT_NAME(const uint8_t *in, size_t len, unsigned bs100k, uint8_t *out, size_t outcap,
                   unsigned nthreads, cpu_mt_blk *blocks, uint32_t *nblocks, double *seconds)
{
  struct cpu_mt_job j;
  struct timespec t0, t1;
  memset(&j, 0, sizeof j);
  j.in = in; j.len = len; j.mbs = (size_t)bs100k * 100000u;
  j.nslabs = (len + j.mbs - 1) / j.mbs;
  pthread_mutex_init(&j.mu, NULL);
  j.sout = calloc(j.nslabs + 1, sizeof *j.sout);
  j.slen = calloc(j.nslabs + 1, sizeof *j.slen);
  j.sblk = calloc(j.nslabs + 1, sizeof *j.sblk);
  j.snb = calloc(j.nslabs + 1, 1);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > j.nslabs && j.nslabs) nthreads = (unsigned)j.nslabs;
  pthread_t *th = calloc(nthreads, sizeof *th);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, cpu_mt_worker, &j);
  for (unsigned i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  /* in-order mux, CRC fold (encode.h:38), header and trailer (compress.c:291-321) */
  size_t o = 0;
  uint32_t combined = 0, nb = 0;
  int ok = outcap >= 14;
  if (ok) { out[o++] = 'B'; out[o++] = 'Z'; out[o++] = 'h'; out[o++] = (uint8_t)('0' + bs100k); }
  for (size_t s = 0; s < j.nslabs; s++) {
    if (ok && o + j.slen[s] + 10 > outcap) ok = 0;
    if (ok) { memcpy(out + o, j.sout[s], j.slen[s]); o += j.slen[s]; }
    for (unsigned b = 0; b < j.snb[s]; b++) {
      const uint32_t c = j.sblk[s][b].crc;
      combined = ((combined << 1) | (combined >> 31)) ^ c ^ 0xFFFFFFFFu;
      if (blocks) { blocks[nb] = j.sblk[s][b]; blocks[nb].slab = (uint32_t)s; blocks[nb].pad_ = 0; }
      nb++;
    }
    free(j.sout[s]);
  }
  if (ok) {
    out[o++] = 0x17; out[o++] = 0x72; out[o++] = 0x45; out[o++] = 0x38; out[o++] = 0x50; out[o++] = 0x90;
    out[o++] = (uint8_t)(combined >> 24); out[o++] = (uint8_t)(combined >> 16);
    out[o++] = (uint8_t)(combined >> 8); out[o++] = (uint8_t)combined;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  if (nblocks) *nblocks = nb;
  free(th); free(j.sout); free(j.slen); free(j.sblk); free(j.snb);
  pthread_mutex_destroy(&j.mu);
  return ok ? o : 0;
}
