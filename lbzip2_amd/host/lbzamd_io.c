/*
 * lbzamd_io.c -- host splitter / muxer around the batch interface: see lbzamd_io.h.
 *
 * Mirrors, for GPU-sized work units, what the reference's process.c does around its block codec: source thread
 * (process.c:260-307), workers (compress.c:73-118), sink thread with the reordering of compress.c:238-250, stream header and
 * trailer (compress.c:291-321).  The unit here is a CHUNK of `chunk_slabs` slabs -- hundreds of blocks, one call of
 * lbzamd_compress_host_body -- not a block, and neither end is a single thread when the file is a regular one.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "../../include/lbzip2_amd.h"
#include "lbzamd_io.h"

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

enum { S_FREE = 0, S_READING, S_FULL, S_BUSY, S_DONE, S_WRITING, S_UNBORN };

struct slot {                       /* one position of the ring: a chunk's input and its compressed bytes, both page-locked */
  uint8_t *in, *out;
  const uint8_t *src;               /* where the chunk's bytes are: `in`, or the chunk's place in the mapped input file */
  size_t len, out_len;
  int state;
  uint64_t seq;
  uint64_t turn;                    /* the chunk this position takes next: position i holds chunks i, i + nslots, i + 2 nslots ... in that order */
};
struct chunk_note {                 /* what the muxer keeps of every chunk: 32 bytes, not its bytes */
  uint64_t off, out_len;
  lbzamd_part part;
  int done;
};
struct engine {
  struct lbzamd_io_cfg cfg;
  int fd_in, fd_out, in_seek, out_seek;
  off_t in_base, out_base;
  int plain_out;
  const uint8_t *map;               /* the input file mapped (a regular file): chunks are ranges of it, nothing is read or page-locked */
  uint64_t map_size;
  int no_populate;                  /* LBZAMD_IO_NOPOPULATE: the mapped chunks are not faulted in ahead of the pipelines (tuning) */
  size_t chunk_bytes, out_cap;
  unsigned nslots, npipes, nreaders, nwriters, next_pipe_id, ctx_ready;
  volatile int watch_stop;
  struct slot *slots;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  uint64_t next_read;               /* the chunk the next reader takes */
  uint64_t total;                   /* chunks in all; UINT64_MAX until the end of the input has been seen */
  uint64_t next_compute;            /* the chunk the next pipeline takes: in order, so offsets become known early */
  uint64_t next_off_seq, next_off;  /* chunks below next_off_seq have their place in the stream; the next one starts at next_off */
  uint64_t next_write, written;
  uint32_t cc;                      /* the stream CRC over the chunks below next_off_seq */
  struct chunk_note *note;
  size_t note_cap;
  int failed, sys_errno;
  char msg[256];
  double t0, t_setup, t_ring, busy_r, busy_w, busy_p;
  uint64_t in_bytes;
};

static void fail_locked(struct engine *e, int kind, int err, const char *msg)
{
  if (!e->failed) {
    e->failed = kind;
    e->sys_errno = err;
    snprintf(e->msg, sizeof e->msg, "%s", msg ? msg : "");
  }
  pthread_cond_broadcast(&e->cv);
}

static struct chunk_note *note_of(struct engine *e, uint64_t seq)       /* mutex held */
{
  if (seq >= e->note_cap) {
    size_t cap = e->note_cap ? e->note_cap * 2u : 64u;
    while (cap <= seq) cap *= 2u;
    struct chunk_note *n = realloc(e->note, cap * sizeof *n);
    if (!n) { fail_locked(e, LBZAMD_IO_MEMORY, ENOMEM, "chunk table"); return NULL; }
    memset(n + e->note_cap, 0, (cap - e->note_cap) * sizeof *n);
    e->note = n;
    e->note_cap = cap;
  }
  return &e->note[seq];
}

/* read(2) / pread(2) until the buffer is full or the input ends; -1 on error */
static ssize_t read_fully(int fd, uint8_t *buf, size_t want, int positioned, off_t at)
{
  size_t n = 0;
  while (n < want) {
    const ssize_t r = positioned ? pread(fd, buf + n, want - n, at + (off_t)n) : read(fd, buf + n, want - n);
    if (r < 0) { if (errno == EINTR) continue; return -1; }
    if (r == 0) break;
    n += (size_t)r;
  }
  return (ssize_t)n;
}

static int write_fully(int fd, const uint8_t *buf, size_t n, int positioned, off_t at)
{
  while (n) {
    const ssize_t w = positioned ? pwrite(fd, buf, n, at) : write(fd, buf, n);
    if (w < 0) { if (errno == EINTR) continue; return -1; }
    buf += w; n -= (size_t)w; at += w;
  }
  return 0;
}

/* LBZAMD_IO_DEBUG=seconds: what every part of the engine is waiting for, on stderr, every so often (a stuck run shows it) */
static void *watch_main(void *arg)
{
  struct engine *e = arg;
  const char *ev = getenv("LBZAMD_IO_DEBUG");
  const int every = ev && atoi(ev) > 0 ? atoi(ev) : 5;
  for (;;) {
    for (int i = 0; i < every * 10; i++) { usleep(100000); if (e->watch_stop) return NULL; }
    pthread_mutex_lock(&e->mu);
    fprintf(stderr, "lbzamd_io: read %llu of %lld, computed %llu, placed %llu, written %llu / next_write %llu, contexts %u of %u, failed %d; ring:",
            (unsigned long long)e->next_read, e->total == UINT64_MAX ? -1ll : (long long)e->total, (unsigned long long)e->next_compute,
            (unsigned long long)e->next_off_seq, (unsigned long long)e->written, (unsigned long long)e->next_write, e->ctx_ready, e->npipes, e->failed);
    for (unsigned i = 0; i < e->nslots; i++) fprintf(stderr, " %llu:%d", (unsigned long long)e->slots[i].seq, e->slots[i].state);
    fputc('\n', stderr);
    pthread_mutex_unlock(&e->mu);
  }
}

static void *reader_main(void *arg)
{
  struct engine *e = arg;
  for (;;) {
    pthread_mutex_lock(&e->mu);
    if (e->failed || e->next_read >= e->total) { pthread_mutex_unlock(&e->mu); return NULL; }
    const uint64_t seq = e->next_read++;
    struct slot *s = &e->slots[seq % e->nslots];
    /* the chunk nslots in front of this one is still on its way out -- or another reader is waiting for this position with
       an EARLIER chunk (more readers than positions, or positions that are born late): that one goes first, or the chunks
       behind it, which are compressed in order, would wait for a position that can only be freed by their own output */
    while ((s->state != S_FREE || s->turn != seq) && !e->failed) pthread_cond_wait(&e->cv, &e->mu);
    if (e->failed || seq >= e->total) { pthread_mutex_unlock(&e->mu); return NULL; }
    s->state = S_READING;
    s->seq = seq;
    pthread_mutex_unlock(&e->mu);

    const double t = now_s();
    ssize_t n;
    if (e->map) {                                           /* nothing to read: the chunk is where the file is mapped */
      const uint64_t at = seq * (uint64_t)e->chunk_bytes;
      n = at >= e->map_size ? 0 : (ssize_t)(e->map_size - at < e->chunk_bytes ? e->map_size - at : e->chunk_bytes);
      s->src = e->map + at;
#ifdef MADV_POPULATE_READ
      /* the chunk's pages into this process's page table NOW, on this thread, which runs ahead of the pipelines (and beside the
         creation of their contexts): the runtime's staging copy of a pageable chunk then takes no page fault per 4 KB.  A hint:
         an error (an older kernel, a file that shrank) changes nothing.  LBZAMD_IO_NOPOPULATE=1: without. */
      if (n > 0 && !e->no_populate)                          /* (in pieces: one call for a whole chunk holds the address space's lock against the contexts' own mappings) */
        for (size_t o = 0; o < (size_t)n; o += (size_t)8 << 20) {
          const size_t len = (size_t)n - o < ((size_t)8 << 20) ? (size_t)n - o : (size_t)8 << 20;
          if (madvise((void *)(uintptr_t)(e->map + at + o), len, MADV_POPULATE_READ)) break;
        }
#endif
    } else {
      n = read_fully(e->fd_in, s->in, e->chunk_bytes, e->in_seek, e->in_base + (off_t)(seq * e->chunk_bytes));
      s->src = s->in;
    }
    const int err = errno;
    const double dt = now_s() - t;

    pthread_mutex_lock(&e->mu);
    e->busy_r += dt;
    if (n < 0) { s->state = S_FREE; fail_locked(e, LBZAMD_IO_READ, err, "read()"); pthread_mutex_unlock(&e->mu); return NULL; }
    if ((size_t)n < e->chunk_bytes) {                       /* the input ends in this chunk (or in front of it) */
      const uint64_t t_end = n ? seq + 1u : seq;
      if (t_end < e->total) e->total = t_end;
    }
    if (n == 0) s->state = S_FREE;                          /* (behind the end of the input: nothing will ask for this position again) */
    else { s->len = (size_t)n; s->state = S_FULL; e->in_bytes += (uint64_t)n; }
    pthread_cond_broadcast(&e->cv);
    const int last = (size_t)n < e->chunk_bytes;
    pthread_mutex_unlock(&e->mu);
    if (last) return NULL;
  }
}

static void *pipeline_main(void *arg)
{
  struct engine *e = arg;
  pthread_mutex_lock(&e->mu);
  const unsigned id = e->next_pipe_id++;
  pthread_mutex_unlock(&e->mu);
  const int device = e->cfg.ndev ? (int)(id % e->cfg.ndev) : -1;
  lbzamd_ctx *ctx = NULL;
  const unsigned long mbs = e->cfg.level * 100000ul;
  if (lbzamd_create(&ctx, device, e->cfg.level, (unsigned)(e->chunk_bytes / mbs), 0)) {
    pthread_mutex_lock(&e->mu);
    e->ctx_ready++;
    fail_locked(e, LBZAMD_IO_DEVICE, 0, lbzamd_last_error());
    pthread_mutex_unlock(&e->mu);
    return NULL;
  }
  pthread_mutex_lock(&e->mu);
  if (e->t_setup == 0.0) e->t_setup = now_s();
  e->ctx_ready++;
  pthread_cond_broadcast(&e->cv);
  pthread_mutex_unlock(&e->mu);
  for (;;) {
    pthread_mutex_lock(&e->mu);
    struct slot *s = NULL;
    uint64_t seq = 0;
    for (;;) {
      if (e->failed) break;
      seq = e->next_compute;
      if (seq >= e->total) break;
      struct slot *c = &e->slots[seq % e->nslots];
      if (c->state == S_FULL && c->seq == seq) { s = c; break; }
      pthread_cond_wait(&e->cv, &e->mu);
    }
    if (!s) { pthread_mutex_unlock(&e->mu); break; }
    e->next_compute++;
    s->state = S_BUSY;
    pthread_mutex_unlock(&e->mu);

    lbzamd_part part;
    const double t = now_s();
    const int rc = lbzamd_compress_host_body(ctx, s->src, s->len, s->out, e->out_cap, &s->out_len, &part);
    const double dt = now_s() - t;

    pthread_mutex_lock(&e->mu);
    e->busy_p += dt;
    if (rc) { fail_locked(e, LBZAMD_IO_DEVICE, 0, lbzamd_last_error()); pthread_mutex_unlock(&e->mu); break; }
    struct chunk_note *n = note_of(e, seq);
    if (!n) { pthread_mutex_unlock(&e->mu); break; }
    n->out_len = s->out_len;
    n->part = part;
    n->done = 1;
    s->state = S_DONE;
    /* every chunk whose predecessors are all compressed now has its place, and the CRC fold moves on (compress.c:246-247) */
    while (e->next_off_seq < e->note_cap && e->note[e->next_off_seq].done) {
      struct chunk_note *q = &e->note[e->next_off_seq];
      q->off = e->next_off;
      e->next_off += q->out_len;
      e->cc = lbzamd_fold_parts(e->cc, &q->part, 1);
      e->next_off_seq++;
    }
    pthread_cond_broadcast(&e->cv);
    pthread_mutex_unlock(&e->mu);
  }
  lbzamd_destroy(ctx);
  return NULL;
}

static void *writer_main(void *arg)
{
  struct engine *e = arg;
  for (;;) {
    pthread_mutex_lock(&e->mu);
    struct slot *s = NULL;
    for (;;) {
      if (e->failed) break;
      if (e->out_seek) {                                  /* any finished chunk that knows where it goes */
        for (unsigned i = 0; i < e->nslots && !s; i++)
          if (e->slots[i].state == S_DONE && e->slots[i].seq < e->next_off_seq) s = &e->slots[i];
      } else {                                            /* a pipe: in order */
        struct slot *c = &e->slots[e->next_write % e->nslots];
        if (c->state == S_DONE && c->seq == e->next_write) { s = c; e->next_write++; }
      }
      if (s) break;
      if (e->total != UINT64_MAX && e->written >= e->total) break;
      if (e->total != UINT64_MAX && !e->out_seek && e->next_write >= e->total) break;
      pthread_cond_wait(&e->cv, &e->mu);
    }
    if (!s) { pthread_mutex_unlock(&e->mu); return NULL; }
    s->state = S_WRITING;
    const off_t at = e->out_base + (off_t)e->note[s->seq].off;
    pthread_mutex_unlock(&e->mu);

    const double t = now_s();
    const int rc = write_fully(e->fd_out, s->out, s->out_len, e->out_seek, at);
    const int err = errno;
    const double dt = now_s() - t;

    pthread_mutex_lock(&e->mu);
    e->busy_w += dt;
    if (rc) { fail_locked(e, LBZAMD_IO_WRITE, err, "write()"); pthread_mutex_unlock(&e->mu); return NULL; }
    s->state = S_FREE;
    s->turn += e->nslots;
    e->written++;
    pthread_cond_broadcast(&e->cv);
    pthread_mutex_unlock(&e->mu);
  }
}

/* can this descriptor be read / written at offsets?  a regular file, not in append mode; *base = where it stands now */
static int positioned_ok(int fd, int for_write, off_t *base, uint64_t *size)
{
  struct stat sb;
  if (fstat(fd, &sb) || !S_ISREG(sb.st_mode)) return 0;
  const off_t cur = lseek(fd, 0, SEEK_CUR);
  if (cur < 0) return 0;
  if (for_write) {
    const int fl = fcntl(fd, F_GETFL);
    if (fl < 0 || (fl & O_APPEND)) return 0;
  } else {
    if (sb.st_size <= cur) return 0;                      /* empty, or a file whose size says nothing (/proc): read it in order */
    *size = (uint64_t)(sb.st_size - cur);
  }
  *base = cur;
  return 1;
}

static int whole_input(int fd, uint8_t **buf, size_t *len, int *pinned);

static int compress_whole(int fd_in, int fd_out, const struct lbzamd_io_cfg *cfg, struct lbzamd_io_stats *st,
                          int *sys_errno, char *msg, size_t msg_cap)
{
  /* lbzip2 -u (compress.c:129-198): one encoder collects across slab boundaries, so the input is one job */
  uint8_t *in = NULL, *out = NULL;
  size_t len = 0, n = 0;
  int pinned = 0, rc = LBZAMD_IO_OK;
  const double t0 = now_s();
  if (whole_input(fd_in, &in, &len, &pinned)) { *sys_errno = errno; snprintf(msg, msg_cap, "read()"); return errno == ENOMEM ? LBZAMD_IO_MEMORY : LBZAMD_IO_READ; }
  const double t1 = now_s();
  const unsigned long mbs = cfg->level * 100000ul;
  const size_t nslabs = (len + mbs - 1) / mbs, cap = lbzamd_bound(len);
  lbzamd_ctx *ctx = NULL;
  out = lbzamd_pinned_alloc(cap);
  if (!out) { rc = LBZAMD_IO_MEMORY; *sys_errno = ENOMEM; snprintf(msg, msg_cap, "page-locked buffer"); goto done; }
  if (lbzamd_create(&ctx, -1, cfg->level, nslabs ? (unsigned)(nslabs > 1200 ? 1200 : nslabs) : 1u, 0)
      || lbzamd_set_sequential(ctx, cfg->sequential)
      || lbzamd_compress_host(ctx, in, len, out, cap, &n)) {
    rc = LBZAMD_IO_DEVICE;
    snprintf(msg, msg_cap, "%s", lbzamd_last_error());
    goto done;
  }
  const double t2 = now_s();
  if (write_fully(fd_out, out, n, 0, 0)) { rc = LBZAMD_IO_WRITE; *sys_errno = errno; snprintf(msg, msg_cap, "write()"); goto done; }
  if (st) {
    memset(st, 0, sizeof *st);
    st->in_bytes = len; st->out_bytes = n; st->chunks = 1; st->pipelines = 1; st->readers = st->writers = 1; st->devices = 1;
    st->seconds = now_s() - t0; st->reader_busy = t1 - t0; st->pipeline_busy = t2 - t1; st->writer_busy = now_s() - t2;
  }
done:
  if (ctx) lbzamd_destroy(ctx);
  if (out) lbzamd_pinned_free(out);
  if (pinned) lbzamd_pinned_free(in); else free(in);
  return rc;
}

int lbzamd_io_compress(int fd_in, int fd_out, const struct lbzamd_io_cfg *cfg, struct lbzamd_io_stats *st,
                       int *sys_errno, char *msg, size_t msg_cap)
{
  int dummy_errno = 0;
  char dummy_msg[8];
  if (!sys_errno) sys_errno = &dummy_errno;
  if (!msg) { msg = dummy_msg; msg_cap = sizeof dummy_msg; }
  *sys_errno = 0;
  msg[0] = 0;
  if (cfg->sequential) return compress_whole(fd_in, fd_out, cfg, st, sys_errno, msg, msg_cap);

  struct engine e;
  memset(&e, 0, sizeof e);
  e.cfg = *cfg;
  e.fd_in = fd_in;
  e.fd_out = fd_out;
  e.t0 = now_s();
  uint64_t in_size = 0;
  e.in_seek = positioned_ok(fd_in, 0, &e.in_base, &in_size);
  e.out_seek = positioned_ok(fd_out, 1, &e.out_base, NULL);
  const unsigned long mbs = cfg->level * 100000ul;
  /* Defaults: chunks of 256 slabs on two pipelines per device.  What a file waits for before its first byte is compressed
     is PAGE-LOCKED memory -- hipHostMalloc pins 5 GB a second (tests/tools/micro/alloc.hip), slower than the file is read, while
     device memory up to some tens of GB usually costs nothing to get -- so the ring is page-locked position by position
     while the first chunks are already on their way; smaller chunks make the ring cheaper and the rounds less efficient
     (64 slabs on four pipelines: 0.3 s sooner, 4.1 instead of 5.4-6.2 GB/s behind the set-up; profiles/r05_filemode_*.txt). */
  e.npipes = (cfg->pipes ? cfg->pipes : 2u) * (cfg->ndev ? cfg->ndev : 1u);
  unsigned chunk_slabs = cfg->chunk_slabs ? cfg->chunk_slabs : 256u;
  if (e.in_seek && !cfg->chunk_slabs) {                   /* a small file: no bigger contexts than its share of slabs per pipeline */
    const uint64_t nslabs = (in_size + mbs - 1u) / mbs;
    const uint64_t share = (nslabs + e.npipes - 1u) / e.npipes;
    if (share < chunk_slabs) chunk_slabs = share ? (unsigned)share : 1u;
    if (nslabs <= 64u) { chunk_slabs = nslabs ? (unsigned)nslabs : 1u; e.npipes = 1u; }
  }
  e.chunk_bytes = (size_t)chunk_slabs * mbs;
  e.out_cap = lbzamd_bound(e.chunk_bytes);
  e.total = UINT64_MAX;
  if (e.in_seek) e.total = (in_size + e.chunk_bytes - 1u) / e.chunk_bytes;
  if (e.total != UINT64_MAX && e.total < e.npipes) e.npipes = e.total ? (unsigned)e.total : 1u;
  /* A regular input file is MAPPED, not read (round 5): what a file waited for was page-locked memory -- hipHostMalloc pins
     5 GB a second, slower than the file is read -- and most of the ring was input.  A chunk is then a range of the mapping and
     the runtime stages it across the link as it does any pageable buffer: 4.2-5.0 GB/s behind the set-up instead of 5.3-6.7
     from page-locked buffers, but the set-up is 0.1-0.2 s instead of 0.6-0.7, which files of up to some tens of GB care about
     more.  LBZAMD_IO_NOMAP=1: read into page-locked buffers as before (a pipe always is). */
  if (e.in_seek && !getenv("LBZAMD_IO_NOMAP") && in_size > 0 && (e.in_base % (off_t)sysconf(_SC_PAGESIZE)) == 0) {
    void *m = mmap(NULL, (size_t)in_size, PROT_READ, MAP_PRIVATE, fd_in, e.in_base);
    if (m != MAP_FAILED) {
      e.map = m;
      e.map_size = in_size;
      (void)madvise(m, (size_t)in_size, MADV_SEQUENTIAL);
      e.no_populate = getenv("LBZAMD_IO_NOPOPULATE") != NULL;
    }
  }
  e.nreaders = e.map ? 1u : (e.in_seek ? (cfg->readers ? cfg->readers : 4u) : 1u);
  e.nwriters = e.out_seek ? (cfg->writers ? cfg->writers : 2u) : 1u;
  if (e.total != UINT64_MAX && e.nreaders > e.total) e.nreaders = e.total ? (unsigned)e.total : 1u;
  e.nslots = 2u * e.npipes + 2u;                          /* per pipeline: one chunk in the device's hands, one arriving or leaving; + one being read, one being written */
  if (e.total != UINT64_MAX && e.nslots > e.total) e.nslots = e.total ? (unsigned)e.total : 1u;
  e.next_off = HEADER_SIZE;
  pthread_mutex_init(&e.mu, NULL);
  pthread_cond_init(&e.cv, NULL);

  int rc = LBZAMD_IO_OK;
  e.slots = calloc(e.nslots, sizeof *e.slots);
  pthread_t *th = calloc(e.nreaders + e.npipes + e.nwriters, sizeof *th);
  unsigned nth = 0;
  if (!e.slots || !th) { rc = LBZAMD_IO_MEMORY; *sys_errno = ENOMEM; snprintf(msg, msg_cap, "chunk ring"); goto out; }
  {
    const uint8_t hdr[HEADER_SIZE] = { 'B', 'Z', 'h', (uint8_t)('0' + cfg->level) };      /* compress.c:291-302 */
    if (write_fully(fd_out, hdr, HEADER_SIZE, e.out_seek, e.out_base)) { rc = LBZAMD_IO_WRITE; *sys_errno = errno; snprintf(msg, msg_cap, "write()"); goto out; }
  }
  /* The contexts take the longest (gigabytes of device memory each): their threads start first and create them side by side,
     while this thread page-locks the ring, position by position -- a position is S_UNBORN until its buffers exist, and the
     readers, which take chunks in order, begin as soon as the first ones do. */
  for (unsigned i = 0; i < e.nslots; i++) { e.slots[i].state = S_UNBORN; e.slots[i].turn = i; }
  {
    /* a thread that cannot be created (RLIMIT_NPROC, --pipelines=64 on 64 devices) fails the job: the ones that exist see
       `failed` and leave, and only they are joined */
    void *(*const mains[3])(void *) = { pipeline_main, reader_main, writer_main };
    const unsigned counts[3] = { e.npipes, e.nreaders, e.nwriters };
    for (unsigned k = 0; k < 3u && !e.failed; k++)
      for (unsigned i = 0; i < counts[k]; i++) {
        const int err = pthread_create(&th[nth], NULL, mains[k], &e);
        if (err) {
          pthread_mutex_lock(&e.mu);
          fail_locked(&e, LBZAMD_IO_MEMORY, err, "pthread_create");
          pthread_cond_broadcast(&e.cv);
          pthread_mutex_unlock(&e.mu);
          break;
        }
        nth++;
      }
  }
  pthread_t watch;
  const int watched = getenv("LBZAMD_IO_DEBUG") != NULL && pthread_create(&watch, NULL, watch_main, &e) == 0;
  /* (Tried: the contexts first, then the ring beside the running pipelines.  The contexts are then ready after 0.1-0.3 s, but
     every hipHostMalloc that follows holds the runtime's lock against the pipelines' copies and launches: 10^9 bytes 0.70-0.90 s
     instead of 0.66, 3 * 10^9 1.3-1.5 s instead of 1.1.  Ring and contexts side by side it is.) */
  for (unsigned i = 0; i < e.nslots && !e.failed; i++) {
    /* With the input mapped the output positions are plain memory too: nothing is page-locked, the library stages the stream
       through device memory and the runtime's own buffers (a quarter of the input's bytes on text).  10^9 bytes file -> file
       0.47 s against 0.59 with page-locked output positions and 0.79-0.88 with the whole ring page-locked; 3 * 10^9: 0.75 s
       against 1.05 and 1.12-1.22 (profiles/r05_filemode_p.txt).  LBZAMD_IO_PINNED_OUT=1: page-locked output positions. */
    const int plain_out = e.map && getenv("LBZAMD_IO_PINNED_OUT") == NULL;
    uint8_t *in = e.map ? NULL : lbzamd_pinned_alloc(e.chunk_bytes), *outb = plain_out ? malloc(e.out_cap) : lbzamd_pinned_alloc(e.out_cap);
    e.plain_out = plain_out;
    pthread_mutex_lock(&e.mu);
    e.slots[i].in = in;
    e.slots[i].out = outb;
    if ((!in && !e.map) || !outb) {
      const int no_dev = lbzamd_device_count() < 1;
      fail_locked(&e, no_dev ? LBZAMD_IO_DEVICE : LBZAMD_IO_MEMORY, ENOMEM, no_dev ? "no HIP device (this program has no CPU path)" : "page-locked chunk buffers");
      pthread_mutex_unlock(&e.mu);
      break;
    }
    e.slots[i].state = S_FREE;
    pthread_cond_broadcast(&e.cv);
    const int stop = e.failed;
    pthread_mutex_unlock(&e.mu);
    if (stop) break;
  }
  e.t_ring = now_s();
  for (unsigned i = 0; i < nth; i++) pthread_join(th[i], NULL);
  if (watched) { e.watch_stop = 1; pthread_join(watch, NULL); }
  if (!e.failed && e.map) {
    /* the job was sized from the file's length when it was opened: a file that grew meanwhile would be compressed short and
       its original removed (the reference reads to the end of the file) */
    struct stat sb;
    if (fstat(fd_in, &sb) == 0 && (uint64_t)sb.st_size != (uint64_t)e.in_base + e.map_size) {
      pthread_mutex_lock(&e.mu);
      fail_locked(&e, LBZAMD_IO_READ, EIO, "read(): the input file changed its size while it was being compressed");
      pthread_mutex_unlock(&e.mu);
    }
  }
  if (e.failed) { rc = e.failed; *sys_errno = e.sys_errno; snprintf(msg, msg_cap, "%s", e.msg); goto out; }
  {
    const uint32_t cc = e.cc;                                                              /* compress.c:304-321 */
    const uint8_t tr[TRAILER_SIZE] = { 0x17, 0x72, 0x45, 0x38, 0x50, 0x90, (uint8_t)(cc >> 24), (uint8_t)(cc >> 16), (uint8_t)(cc >> 8), (uint8_t)cc };
    if (write_fully(fd_out, tr, TRAILER_SIZE, e.out_seek, e.out_base + (off_t)e.next_off)) { rc = LBZAMD_IO_WRITE; *sys_errno = errno; snprintf(msg, msg_cap, "write()"); goto out; }
    if (e.out_seek) (void)lseek(fd_out, e.out_base + (off_t)e.next_off + TRAILER_SIZE, SEEK_SET);   /* leave the descriptor behind the stream, as write() would */
    if (e.in_seek) (void)lseek(fd_in, e.in_base + (off_t)e.in_bytes, SEEK_SET);
  }
out:;
  const double t1 = now_s();
  if (st) {
    memset(st, 0, sizeof *st);
    st->in_bytes = e.in_bytes;
    st->out_bytes = rc ? 0 : e.next_off + TRAILER_SIZE;
    st->chunks = e.total == UINT64_MAX ? 0 : e.total;
    st->readers = e.nreaders; st->writers = e.nwriters; st->pipelines = e.npipes; st->devices = cfg->ndev ? cfg->ndev : 1u;
    st->chunk_slabs = chunk_slabs;
    st->seconds = t1 - e.t0;
    st->setup_seconds = e.t_setup > 0.0 ? e.t_setup - e.t0 : 0.0;
    st->reader_busy = e.busy_r; st->writer_busy = e.busy_w; st->pipeline_busy = e.busy_p;
  }
  if (cfg->report && !rc) {
    const double w = t1 - e.t0;
    fprintf(stderr, "file splitter/muxer: %llu B -> %llu B in %.3f s = %.0f MB/s (contexts included: first one ready after %.3f s, ring page-locked after %.3f s; %.0f MB/s behind the first context); "
                    "%u pipeline(s) on %u device(s), chunks of %u slabs%s; %u reader(s) busy %.0f%% each, %u writer(s) busy %.0f%% each, pipelines busy %.0f%%\n",
            (unsigned long long)e.in_bytes, (unsigned long long)(e.next_off + TRAILER_SIZE), w, (double)e.in_bytes / w / 1e6,
            e.t_setup > 0.0 ? e.t_setup - e.t0 : 0.0, e.t_ring - e.t0, (double)e.in_bytes / (w - (e.t_setup > 0.0 ? e.t_setup - e.t0 : 0.0)) / 1e6,
            e.npipes, cfg->ndev ? cfg->ndev : 1u, chunk_slabs, e.map ? " of the mapped input" : "",
            e.nreaders, 100.0 * e.busy_r / (w * e.nreaders), e.nwriters, 100.0 * e.busy_w / (w * e.nwriters), 100.0 * e.busy_p / (w * e.npipes));
  }
  if (e.slots) for (unsigned i = 0; i < e.nslots; i++) {
    if (e.slots[i].in) lbzamd_pinned_free(e.slots[i].in);
    if (e.slots[i].out) { if (e.plain_out) free(e.slots[i].out); else lbzamd_pinned_free(e.slots[i].out); }
  }
  if (e.map) (void)munmap((void *)e.map, (size_t)e.map_size);
  free(e.slots);
  free(e.note);
  free(th);
  pthread_mutex_destroy(&e.mu);
  pthread_cond_destroy(&e.cv);
  return rc;
}

/* the whole of fd into one buffer: page-locked when the size is known up front (a regular file), grown otherwise */
static int whole_input(int fd, uint8_t **buf, size_t *len, int *pinned)
{
  off_t base = 0;
  uint64_t size = 0;
  *pinned = 0;
  if (positioned_ok(fd, 0, &base, &size)) {
    uint8_t *b = lbzamd_pinned_alloc((size_t)size + 1u);
    if (b) {
      const ssize_t n = read_fully(fd, b, (size_t)size + 1u, 0, 0);      /* (+1: a file that grew meanwhile shows here) */
      if (n < 0) { const int e = errno; lbzamd_pinned_free(b); errno = e; return -1; }
      if ((uint64_t)n <= size) { *buf = b; *len = (size_t)n; *pinned = 1; return 0; }
      /* it grew: fall through to the growing buffer with what there is */
      size_t cap = (size_t)n * 2u;
      uint8_t *g = malloc(cap);
      if (!g) { lbzamd_pinned_free(b); errno = ENOMEM; return -1; }
      memcpy(g, b, (size_t)n);
      lbzamd_pinned_free(b);
      size_t have = (size_t)n;
      for (;;) {
        if (have == cap) { cap *= 2u; uint8_t *g2 = realloc(g, cap); if (!g2) { free(g); errno = ENOMEM; return -1; } g = g2; }
        const ssize_t r = read_fully(fd, g + have, cap - have, 0, 0);
        if (r < 0) { const int e = errno; free(g); errno = e; return -1; }
        have += (size_t)r;
        if (have < cap) break;
      }
      *buf = g; *len = have;
      return 0;
    }
  }
  size_t cap = 1u << 22, have = 0;
  uint8_t *g = malloc(cap);
  if (!g) { errno = ENOMEM; return -1; }
  for (;;) {
    const ssize_t r = read_fully(fd, g + have, cap - have, 0, 0);
    if (r < 0) { const int e = errno; free(g); errno = e; return -1; }
    have += (size_t)r;
    if (have < cap) break;
    cap *= 2u;
    uint8_t *g2 = realloc(g, cap);
    if (!g2) { free(g); errno = ENOMEM; return -1; }
    g = g2;
  }
  *buf = g; *len = have;
  return 0;
}

/* .bz2 -> bytes in BOUNDED memory (round 6; until then the whole input was read and every block decoded before a byte was
 * written, where the reference reads, decodes and writes as it goes: src/process.c:260-307, src/expand.c:547-690): the input is
 * taken a WINDOW at a time -- LBZAMD_IO_DWINDOW bytes, 256 MB by default --, the whole blocks of the window are decoded at once
 * (lbzamd_decompress_window) and written before more is read; what is left of the window -- from the magic of the first block
 * that may be cut off -- moves to the front of the next one.  `bzcat big.bz2 | head` ends when head does, a stream of any length
 * goes through in the memory of one window and its bytes.  An input that fits one window is decoded exactly as before. */
int lbzamd_io_decompress(int fd_in, int fd_out, int not_bzip2_copy, int report, struct lbzamd_io_stats *st,
                         int *sys_errno, int *err_code, char *msg, size_t msg_cap)
{
  int dummy = 0, dummy2 = 0;
  char dummy_msg[8];
  if (!sys_errno) sys_errno = &dummy;
  if (!err_code) err_code = &dummy2;
  if (!msg) { msg = dummy_msg; msg_cap = sizeof dummy_msg; }
  *sys_errno = 0; *err_code = 0; msg[0] = 0;
  size_t cap = 256u << 20;
  { const char *e = getenv("LBZAMD_IO_DWINDOW"); if (e && atol(e) > 0) cap = (size_t)atol(e); }
  if (cap < 64u) cap = 64u;
  {
    off_t base = 0; uint64_t size = 0;
    if (positioned_ok(fd_in, 0, &base, &size) && size + 1u < cap) cap = (size_t)size + 1u;      /* a short file: no more than it needs (+1: its end shows) */
  }
  uint8_t *buf = malloc(cap), *out = NULL;
  size_t have = 0, n = 0, zlen = 0, total = 0;
  int rc = LBZAMD_IO_OK, eof = 0, first = 1;
  unsigned windows = 0;
  lbzamd_dctx *d = NULL;
  lbzamd_dresume rs;
  lbzamd_dstats acc;
  memset(&rs, 0, sizeof rs);
  memset(&acc, 0, sizeof acc);
  const double t0 = now_s();
  double t_read = 0, t_dec = 0, t_write = 0;
  if (!buf) { *sys_errno = ENOMEM; snprintf(msg, msg_cap, "read()"); return LBZAMD_IO_MEMORY; }
  for (;;) {
    double t = now_s();
    while (have < cap && !eof) {
      const ssize_t r = read(fd_in, buf + have, cap - have);
      if (r < 0) { if (errno == EINTR) continue; rc = LBZAMD_IO_READ; *sys_errno = errno; snprintf(msg, msg_cap, "read()"); goto done; }
      if (r == 0) eof = 1; else { have += (size_t)r; zlen += (size_t)r; }
    }
    t_read += now_s() - t;
    if (first) {
      /* process.c:664-681: four bytes decide whether this is a bzip2 file at all */
      const int is_bz = have >= 4 && buf[0] == 'B' && buf[1] == 'Z' && buf[2] == 'h' && buf[3] >= '1' && buf[3] <= '9';
      if (!is_bz) {
        if (not_bzip2_copy && fd_out >= 0) {                  /* copied through as it is, a window at a time */
          for (;;) {
            if (have && write_fully(fd_out, buf, have, 0, 0)) { rc = LBZAMD_IO_WRITE; *sys_errno = errno; snprintf(msg, msg_cap, "write()"); goto done; }
            total += have; have = 0;
            if (eof) break;
            const ssize_t r = read(fd_in, buf, cap);
            if (r < 0) { if (errno == EINTR) continue; rc = LBZAMD_IO_READ; *sys_errno = errno; snprintf(msg, msg_cap, "read()"); goto done; }
            if (r == 0) eof = 1; else { have = (size_t)r; zlen += (size_t)r; }
          }
          goto done;
        }
        rc = LBZAMD_IO_DATA; *err_code = 3;                   /* ERR_MAGIC */
        goto done;
      }
      unsigned maxb = (unsigned)(have / 20000u + 8u);
      if (lbzamd_dcreate(&d, -1, maxb > 2400u ? 2400u : maxb)) { rc = LBZAMD_IO_DEVICE; snprintf(msg, msg_cap, "%s", lbzamd_last_error()); goto done; }
      first = 0;
    }
    if (rs.finished) {                                        /* behind the last stream: read to the end, as the reference does, and ignore it */
      have = 0;
      if (eof) break;
      continue;
    }
    t = now_s();
    const int drc = lbzamd_decompress_window(d, buf, have, eof, &rs, &out, &n);
    t_dec += now_s() - t;
    windows++;
    {
      lbzamd_dstats ds;
      lbzamd_dget_stats(d, &ds);
      acc.nblocks += ds.nblocks; acc.nstreams += ds.nstreams; acc.ms_scan += ds.ms_scan; acc.ms_blocks += ds.ms_blocks; acc.ms_emit += ds.ms_emit;
    }
    if (drc == -3) {
      rc = LBZAMD_IO_DATA; *err_code = lbzamd_last_error_code(); snprintf(msg, msg_cap, "%s", lbzamd_last_error());
      /* the bytes in front of the damage -- the whole blocks before the one that was refused -- go out as the reference's do
         (it has written what it decoded by then); a caller that writes a FILE removes it, as lbzip2 does */
      if (fd_out >= 0 && out && n) (void)write_fully(fd_out, out, n, 0, 0);
      goto done;
    }
    if (drc) { rc = LBZAMD_IO_DEVICE; snprintf(msg, msg_cap, "%s", lbzamd_last_error()); goto done; }
    t = now_s();
    if (fd_out >= 0 && n && write_fully(fd_out, out, n, 0, 0)) { rc = LBZAMD_IO_WRITE; *sys_errno = errno; snprintf(msg, msg_cap, "write()"); goto done; }
    t_write += now_s() - t;
    total += n;
    lbzamd_free(out); out = NULL; n = 0;
    if (eof) break;
    const size_t keep = (size_t)(rs.consumed_bit / 8u);
    if (keep == 0) {                                          /* no whole block in the window: a larger one */
      uint8_t *b2 = realloc(buf, cap * 2u);
      if (!b2) { rc = LBZAMD_IO_MEMORY; *sys_errno = ENOMEM; snprintf(msg, msg_cap, "read()"); goto done; }
      buf = b2; cap *= 2u;
    } else {
      memmove(buf, buf + keep, have - keep);
      have -= keep;
    }
  }
done:;
  const double t3 = now_s();
  if (st) {
    memset(st, 0, sizeof *st);
    st->in_bytes = zlen; st->out_bytes = rc ? 0 : total; st->chunks = windows ? windows : 1; st->pipelines = 1; st->readers = st->writers = 1; st->devices = 1;
    st->seconds = t3 - t0; st->reader_busy = t_read; st->pipeline_busy = t_dec; st->writer_busy = t_write;
  }
  if (report && !rc && d) {
    fprintf(stderr, "decode: %zu B -> %zu B, %u blocks in %u stream(s), %u window(s); read %.3f s, context + decode %.3f s = %.0f MB/s (device: scan %.1f blocks %.1f emit %.1f ms), write %.3f s\n",
            zlen, total, acc.nblocks, acc.nstreams, windows, t_read, t_dec, (double)total / (t_dec > 0 ? t_dec : 1e-9) / 1e6, acc.ms_scan, acc.ms_blocks, acc.ms_emit, t_write);
  }
  if (d) lbzamd_ddestroy(d);
  lbzamd_free(out);
  free(buf);
  return rc;
}
