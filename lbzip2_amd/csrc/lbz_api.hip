/*
 * lbz_api.hip -- host runtime and C ABI (include/lbzip2_amd.h) of the MI355X block
 * compressor: device buffers laid out per block (lbz_common.h), rounds of blocks pipelined
 * over two HIP streams per context (run_chunk), HIP events around every kernel, chunked
 * streaming for inputs larger than the resident capacity, and the drop-in encode.h work-unit
 * functions on top of a pool of one-slab contexts.
 *
 * Mirrors the call sequence of the reference's src/compress.c (work units :73-118,
 * transmit :210-228, reorder + CRC fold :238-250, header/trailer :291-321); nothing here
 * computes any part of the codec on the CPU.
 */
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lbzip2_amd.h"
#include "lbz_kernels.h"

static thread_local std::string g_err;
static int fail_msg(const char *what, hipError_t e)
{
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  g_err = buf;
  return -1;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail_msg(#x, e_); } while (0)

extern "C" const char *lbzamd_last_error(void) { return g_err.c_str(); }
/* the reference's enum error (src/common.h:54-76) for a stream the decoder refused (return value -3), 0 otherwise */
static thread_local int g_err_code;
extern "C" int lbzamd_last_error_code(void) { return g_err_code; }
enum { RE_MAGIC = 3, RE_HEADER = 4, RE_BLKCRC = 15, RE_STRMCRC = 16, RE_EOF = 19 };
static int dec_error(int code, uint32_t nblock);

static inline uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1u) / a * a; }

struct lbzamd_ctx {
  int device = 0;
  unsigned bs100k = 9;
  lbz_layout L{};
  uint32_t max_slabs = 0, nslots = 0, ncus = 0;
  uint64_t slot_bytes = 0, spill_bytes = 0;      /* BWT workspace of a full-size / a spill block */
  hipStream_t stream = nullptr;               /* everything a caller can observe happens in order on this one */
  hipStream_t side[7] = {};                   /* rounds of a chunk go round-robin over stream + side[0 .. nstreams-2] */
  unsigned nstreams = 2;
  hipEvent_t ev[8] = {};
  std::vector<hipEvent_t> bev;                /* a start/end pair around every per-round launch of a chunk */
  std::vector<hipEvent_t> jev;                /* side streams -> caller's stream joins */
  std::vector<hipEvent_t> fev;                /* a round's stream offsets are known: the next round's may follow */
  /* host-buffer calls: every round's slabs come in on ONE copy stream, in round order at the full rate of the link (copies
     on several streams share it and all arrive late); a short head round on a lane of its own starts the device early */
  hipStream_t copy_q = nullptr, head_q = nullptr;
  std::vector<hipEvent_t> cev;                /* round i's slabs are in device memory */
  hipEvent_t head_ev = nullptr;
  u8 *ws_head = nullptr;                      /* BWT workspaces of the head round */
  std::vector<int> bkind;                     /* which kernel each pair times (index into kms) */
  float kms[6] = { 0, 0, 0, 0, 0, 0 };       /* partition, batch, fix, mtf, encode, collect: accumulated per call */
  /* device */
  u8 *T = nullptr, *B = nullptr, *R = nullptr, *O = nullptr, *ws = nullptr;
  u16 *V = nullptr;
  u32 *freq = nullptr;
  u64 *offs = nullptr;
  lbz_block_meta *meta = nullptr;
  lbz_stream_state *st = nullptr;
  u8 *d_in = nullptr, *d_out = nullptr;      /* staging for the host-buffer path */
  const u8 *h2d_host = nullptr;              /* host-buffer call in progress: rounds copy their own slabs in */
  bool sequential = false;                   /* -u: blocks span slab boundaries (k_collect_seq) */
  bool out_is_host = false;                  /* host-buffer call in progress whose output is page-locked: the device writes it */
  unsigned long long *seq_starts = nullptr;  /* max_slabs + 1 chain entries, ticket, lbz_seq_out */
  u32 *seq_ticket = nullptr;
  lbz_seq_out *seq_out = nullptr;
  u32 *seq_tab32 = nullptr;                  /* step tables of the sequential mode: last head, first head, emitted bytes per step */
  unsigned long long *seq_tab64 = nullptr;   /* run start carried into each step, prefix sums of the emitted bytes */
  size_t seq_tab_cap = 0;                    /* steps the tables hold */
  size_t d_in_cap = 0, d_out_cap = 0;
  /* host */
  std::vector<lbz_block_meta> h_meta;
  uint32_t last_nslabs = 0;
  size_t nbev_used = 0;
  lbzamd_stats stats{};
};

static int ctx_free(lbzamd_ctx *c)
{
  if (!c) return 0;
  (void)hipFree(c->T); (void)hipFree(c->B); (void)hipFree(c->R); (void)hipFree(c->O); (void)hipFree(c->ws); (void)hipFree(c->V);
  (void)hipFree(c->freq); (void)hipFree(c->offs); (void)hipFree(c->meta); (void)hipFree(c->st);
  (void)hipFree(c->d_in); (void)hipFree(c->d_out);
  (void)hipFree(c->seq_starts); (void)hipFree(c->seq_ticket); (void)hipFree(c->seq_out);
  (void)hipFree(c->seq_tab32); (void)hipFree(c->seq_tab64);
  for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->bev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->jev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->fev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->cev) if (e) (void)hipEventDestroy(e);
  if (c->head_ev) (void)hipEventDestroy(c->head_ev);
  if (c->copy_q) (void)hipStreamDestroy(c->copy_q);
  if (c->head_q) (void)hipStreamDestroy(c->head_q);
  (void)hipFree(c->ws_head);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  for (auto &q : c->side) if (q) (void)hipStreamDestroy(q);
  delete c;
  return 0;
}

extern "C" void lbzamd_destroy(lbzamd_ctx *c) { ctx_free(c); }

/* Devices as the callers count them.  LBZAMD_FAKE_DEVICES=N shows N logical devices whatever the box holds, logical device
 * i living on physical device i mod (devices present): every N > 1 branch of the host side -- contexts dealt over devices
 * (lbzamd_compress -g), one work-unit pool per device (LBZAMD_DEVICES), lbzamd_device_count -- runs on the real kernels
 * of a one-GPU box that way (tests/test_gpu_parity.py: test_n_devices_on_one_gpu).  Unset: the devices present.   */
static int physical_devices()
{
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
static int logical_devices()
{
  const int have = physical_devices();
  const char *env = getenv("LBZAMD_FAKE_DEVICES");
  const int fake = env ? atoi(env) : 0;
  return have < 1 ? 0 : (fake > 0 ? (fake > 64 ? 64 : fake) : have);
}
static int physical_of(int logical)
{
  const int have = physical_devices();
  return have > 0 ? logical % have : 0;
}

static int ctx_create(lbzamd_ctx **out, int device, unsigned bs100k, unsigned max_slabs, unsigned nslots, unsigned force_streams);

extern "C" int lbzamd_create(lbzamd_ctx **out, int device, unsigned bs100k, unsigned max_slabs, unsigned nslots)
{
  return ctx_create(out, device, bs100k, max_slabs, nslots, 0u);
}

/* force_streams != 0: that many round streams (and slot sets) whatever the environment says -- the
 * work-unit pool runs its rounds on one stream and must not pay for a second set of BWT workspaces */
static int ctx_create(lbzamd_ctx **out, int device, unsigned bs100k, unsigned max_slabs, unsigned nslots, unsigned force_streams)
{
  if (!out || bs100k < 1 || bs100k > 9 || max_slabs < 1) { g_err = "lbzamd_create: bad argument"; return -1; }
  const int ndev = logical_devices();
  if (ndev < 1) { g_err = "lbzamd_create: no HIP device (this library has no CPU path)"; return -1; }
  if (device < 0) HIPCHK(hipGetDevice(&device));
  if (device >= ndev) { g_err = "lbzamd_create: no such device"; return -1; }
  device = physical_of(device);                  /* from here on the physical device (LBZAMD_FAKE_DEVICES) */
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));

  lbzamd_ctx *c = new lbzamd_ctx;
  c->device = device;
  c->bs100k = bs100k;
  c->max_slabs = max_slabs;
  c->ncus = (uint32_t)prop.multiProcessorCount;
  const uint32_t M = bs100k * 100000u;
  c->L.M = M;
  c->L.cap_a = round_up(M + 64u, 256u);
  c->L.cap_b = round_up(M / 4u + 128u, 256u);
  c->L.out_a = round_up(M + M / 8u + 4096u, 256u);
  c->L.out_b = round_up(c->L.cap_b + c->L.cap_b / 8u + 4096u, 256u);
  c->slot_bytes = (LBZ_BWT_SLOT_BYTES(c->L.cap_a) + 255u) & ~(uint64_t)255u;
  c->spill_bytes = (LBZ_BWT_SLOT_BYTES(c->L.cap_b) + 255u) & ~(uint64_t)255u;
  {
    const char *env = getenv("LBZAMD_STREAMS");
    c->nstreams = env ? (unsigned)atoi(env) : 3u;
    if (c->nstreams < 1u) c->nstreams = 1u;
    if (c->nstreams > 8u) c->nstreams = 8u;
    if (force_streams) c->nstreams = force_streams;
  }
  if (nslots == 0) {
    /* Slabs per round: the chunk's slabs dealt evenly over the streams (rounds of equal size overlap best), as far as
       half of the free device memory allows (a slot is 48 B per block byte, + 1/4 for the spill).  A round of a few
       dozen blocks already fills the sorting kernels -- from k_bwt_batch on a block is LBZ_BWT_SEGS workgroups. */
    const char *env = getenv("LBZAMD_SLOTS");
    const unsigned floor_slots = (unsigned)prop.multiProcessorCount;    /* up to one block per CU goes as ONE round: the partition, MTF and
                                                                           coding kernels are a workgroup per block, and a block's chain of
                                                                           launches is what a small input waits for (two rounds of 56 blocks
                                                                           on two streams: 26.5 ms for 10^8 bytes, one round: see DESIGN.md 6) */
    if (env) {
      nslots = (unsigned)atoi(env);
    } else {
      nslots = (max_slabs + c->nstreams - 1u) / c->nstreams;
      if (nslots < floor_slots) nslots = floor_slots;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t fit = free_b / 2u / ((size_t)c->nstreams * (c->slot_bytes + c->spill_bytes));
        if (nslots > fit) nslots = fit > floor_slots ? (unsigned)fit : floor_slots;
      }
    }
    if (nslots == 0) nslots = 1;
  }
  if (nslots > max_slabs) nslots = max_slabs;
  c->nslots = nslots;
  if (max_slabs <= nslots) c->nstreams = 1u;                 /* a single round: nothing to overlap */

  const size_t elems = (size_t)max_slabs * ((size_t)c->L.cap_a + c->L.cap_b);
  const size_t outb = (size_t)max_slabs * ((size_t)c->L.out_a + c->L.out_b);
  const size_t nblk = 2u * (size_t)max_slabs;
#define ALLOC(p, bytes) do { hipError_t e_ = hipMalloc((void **)&(p), (bytes)); \
    if (e_ != hipSuccess) { ctx_free(c); return fail_msg("hipMalloc " #p, e_); } } while (0)
  ALLOC(c->T, elems);
  ALLOC(c->B, elems);
  ALLOC(c->R, elems);
  ALLOC(c->V, elems * 2u);
  ALLOC(c->O, outb);
  ALLOC(c->ws, (size_t)c->nstreams * nslots * (c->slot_bytes + c->spill_bytes));   /* one set of slots per stream */
  ALLOC(c->freq, nblk * 260u * sizeof(u32));
  ALLOC(c->offs, nblk * sizeof(u64));
  ALLOC(c->meta, nblk * sizeof(lbz_block_meta));
  ALLOC(c->st, sizeof(lbz_stream_state));
#undef ALLOC
  hipError_t e = hipStreamCreate(&c->stream);
  if (e != hipSuccess) { ctx_free(c); return fail_msg("hipStreamCreate", e); }
  for (unsigned i = 0; i + 1 < c->nstreams; i++) {
    e = hipStreamCreate(&c->side[i]);
    if (e != hipSuccess) { ctx_free(c); return fail_msg("hipStreamCreate", e); }
  }
  for (auto &ev : c->ev) {
    e = hipEventCreate(&ev);
    if (e != hipSuccess) { ctx_free(c); return fail_msg("hipEventCreate", e); }
  }
  c->h_meta.resize(nblk);
  *out = c;
  return 0;
}

extern "C" size_t lbzamd_bound(size_t len)
{
  return len + len / 8u + 8192u + (len / 100000u + 2u) * 64u;
}

extern "C" void *lbzamd_stream(lbzamd_ctx *c) { return (void *)c->stream; }
extern "C" uint32_t lbzamd_slots(lbzamd_ctx *c) { return c ? c->nslots : 0u; }

/* Enqueue stages [0, upto] for one chunk of nsl slabs already resident at d_in.
 *
 * The chunk's slabs are cut into rounds of nslots.  A round is a chain of six launches -- RLE1 +
 * CRC, partition, batches, deep ties, MTF, prefix codes + packing -- with one workgroup
 * per block: the primary blocks of the round's slabs first, then their (usually empty) spill
 * blocks; through the three BWT kernels a workgroup owns one workspace slot.  With two streams,
 * rounds alternate between them, each stream with its own set of slots: a round's launches stay
 * ordered, but workgroups of the other stream's round fill the CUs that a kernel boundary, or
 * a round with fewer blocks than CUs, would leave idle.  A last round shorter than the others
 * is issued first so that the chunk does not END on a half-empty device.
 * The caller's stream joins the side stream before anything else is enqueued on it.          */
static int timed_begin(lbzamd_ctx *c, size_t *nbev, int kind, hipStream_t s)
{
  while (c->bev.size() < *nbev + 2) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    c->bev.push_back(e);
  }
  if (c->bkind.size() < c->bev.size() / 2) c->bkind.resize(c->bev.size() / 2);
  c->bkind[*nbev / 2] = kind;
  HIPCHK(hipEventRecord(c->bev[*nbev], s));
  return 0;
}
static int timed_end(lbzamd_ctx *c, size_t *nbev, hipStream_t s)
{
  HIPCHK(hipEventRecord(c->bev[*nbev + 1], s));
  *nbev += 2;
  return 0;
}

/* Workgroups per block of a round of `count` blocks: in the sorting kernels (segments), and in the partition (1 = k_bwt_part,
 * every pass in one launch; else the launch-per-pass kernels with that many workgroups per block).  LBZAMD_PARTS: (tuning). */
static u32 round_segs(const lbzamd_ctx *c, u32 count)
{
  static const int forced = getenv("LBZAMD_SEGS") ? atoi(getenv("LBZAMD_SEGS")) : 0;      /* (tuning) */
  if (forced > 0 && forced <= (int)LBZ_BWT_MAXSEGS) return (u32)forced;
  return count <= c->ncus ? LBZ_BWT_MAXSEGS : LBZ_BWT_SEGS;
}
static u32 round_parts(const lbzamd_ctx *c, u32 count, bool overlapped)
{
  static const int forced = getenv("LBZAMD_PARTS") ? atoi(getenv("LBZAMD_PARTS")) : 0;
  if (forced > 0 && forced <= 16) return (u32)forced == 1u ? 1u : (u32)forced;
  if (overlapped && count > c->ncus) return 1u;
  return count <= c->ncus / 2u ? 16u : 8u;
}

extern "C" void lbzamd_round_shape(lbzamd_ctx *c, uint32_t blocks, int overlapped, uint32_t *segments, uint32_t *partition_wgs)
{
  if (segments) *segments = c ? round_segs(c, blocks) : 0u;
  if (partition_wgs) *partition_wgs = c ? round_parts(c, blocks, overlapped != 0) : 0u;
}

/* The sorter's launches for the blocks of one round (nblk = 2 * count: primaries then spills; or count: the
 * listed primaries only): partition (one workgroup per block), then batches, tie lists, the deep-tie rounds --
 * one launch per doubling depth, every (block, segment) a workgroup (k_bwt.hip) -- and the origin pointers.     */
/* Move-to-front ranks, zero runs and the histogram of a round's blocks: one workgroup per block in one launch when the round
 * fills the device by itself; for rounds of fewer blocks than CUs the ranks -- nine tenths of the kernel -- are dealt over 2 or
 * 4 workgroups per block (k_mtf_ranks) and the zero-run coding follows in a launch of its own (k_mtf_zrle).  `real`: blocks
 * that hold bytes, about (the spill blocks of a round are mostly empty).  LBZAMD_MTF_PARTS: (tuning) 1, 2 or 4. */
static void launch_mtf(lbzamd_ctx *c, hipStream_t q, u32 first, u32 count, u32 nblk, u32 real, const u32 *lst)
{
  const char *e = getenv("LBZAMD_MTF_PARTS");
  u32 parts = e ? (u32)atoi(e) : (real <= c->ncus / 2u ? 4u : (real <= c->ncus ? 2u : 1u));
  if (parts != 2u && parts != 4u) parts = 1u;
  if (parts == 1u) {
    hipLaunchKernelGGL(k_mtf, dim3(nblk), dim3(LBZ_MTF_WG), 0, q, (const u8 *)c->B, c->R, c->V, c->freq, c->meta, c->L, first, count, lst);
    return;
  }
  hipLaunchKernelGGL(k_mtf_ranks, dim3(nblk * parts), dim3(LBZ_MTF_WG), 0, q, (const u8 *)c->B, c->R, c->meta, c->L, first, count, lst, parts);
  hipLaunchKernelGGL(k_mtf_zrle, dim3(nblk), dim3(LBZ_MTF_WG), 0, q, (const u8 *)c->R, c->V, c->freq, c->meta, c->L, first, count, lst);
}

static void launch_sort(lbzamd_ctx *c, hipStream_t q, u32 first, u32 count, u32 nblk, u8 *ws, u8 *wsp, const u32 *lst,
                        int phase /* 0 = partition, 1 = batches, 2 = deep ties (3: the text rounds alone, 4: the rank rounds alone) */, bool overlapped = false /* other rounds run beside this one */)
{
  /* workgroups per block in the sorting kernels: more of them when the round has fewer blocks than the device has CUs -- the
     caller then waits for a block's chain of launches, and every launch is as long as its longest segment */
  const u32 segs = round_segs(c, count);
  if (phase == 0 && round_parts(c, count, overlapped) == 1u) {
    /* big rounds side by side on several streams: one workgroup per block, every pass in one launch.  Three rounds of 371
       blocks overlapped: 8.1 GB/s against 7.8 with the launch-per-pass form, which alone on the device is the faster one
       (371 blocks: 9.6 ms against 14; 1112 blocks: 28.0 against 29.9) */
    hipLaunchKernelGGL(k_bwt_part, dim3(nblk), dim3(LBZ_BWT_WG), 0, q, (const u8 *)c->T, c->meta, c->L,
                       first, count, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst, segs);
  } else if (phase == 0) {
    /* the partition, a launch per pass: more workgroups per block the fewer blocks the round has (a block's passes are what
       a small input waits for; a full device needs only enough workgroups to fill it evenly) */
    const u32 parts = round_parts(c, count, overlapped);
    const dim3 g(lbz_seg_grid(nblk, parts));
    hipLaunchKernelGGL(k_bwt_hist, g, dim3(LBZ_BWT_WG), 0, q, (const u8 *)c->T, c->meta, c->L,
                       first, count, nblk, parts, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst);
    for (u32 pass = 0; pass < 4u; pass++)
      hipLaunchKernelGGL(k_bwt_scat, g, dim3(LBZ_BWT_WG), 0, q, (const u8 *)c->T, c->meta, c->L,
                         first, count, nblk, parts, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst, pass);
    hipLaunchKernelGGL(k_bwt_segs, dim3(nblk), dim3(LBZ_BWT_WG), 0, q, c->meta, c->L,
                       first, count, segs, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst);
  } else if (phase == 1) {
    hipLaunchKernelGGL(k_bwt_batch, dim3(lbz_seg_grid(nblk, segs)), dim3(LBZ_BWT_WG), 0, q, (const u8 *)c->T, c->B, c->meta, c->L,
                       first, count, nblk, segs, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst);
  } else {
    static const unsigned deep_pad = getenv("LBZAMD_DEEP_PAD") ? (unsigned)atoi(getenv("LBZAMD_DEEP_PAD")) : 0u;   /* (tuning) idle LDS: fewer workgroups per CU */
    /* when a block goes to the rank rounds, in thousandths of its rows tied: as the text rounds begin (0 = no such rule) and
       after their first launch (k_bwt_deep) */
    const char *e0 = getenv("LBZAMD_HANDOVER0"), *e1 = getenv("LBZAMD_HANDOVER1");   /* (tuning; read per round) */
    const u32 ho0 = e0 ? (u32)atoi(e0) : LBZ_HANDOVER0;
    const u32 ho1 = e1 ? (u32)atoi(e1) : LBZ_HANDOVER1;
    const u32 handover = (ho0 & 0xFFFFu) | (ho1 << 16);
    const dim3 g(lbz_seg_grid(nblk, segs));
    const u32 R = lbz_fix_rounds(c->L.M);
    for (u32 r = 0; r < LBZ_DEEP_ROUNDS && phase != 4; r++)
      if (r <= LBZ_DEEP_BUILD)
        hipLaunchKernelGGL(k_bwt_deep, g, dim3(LBZ_BWT_WG), deep_pad, q, (const u8 *)c->T, c->B, c->meta, c->L,
                           first, count, nblk, segs, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst, r, handover);
      else                                               /* the launches that may step by ranks */
        hipLaunchKernelGGL(k_bwt_deepr, g, dim3(LBZ_BWT_WG), deep_pad, q, (const u8 *)c->T, c->B, c->meta, c->L,
                           first, count, nblk, segs, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst, r, handover);
    /* ONE chain of rank rounds behind the last text launch, for every block with ties left (which = 0).  Round 5 also ran the
       blocks handed over early on a stream of their own beside the later text launches (LBZAMD_SPLIT_CHAIN): a round alone on
       the device lost a quarter of its tie stages' time that way, three overlapping rounds gained nothing and short rounds
       lost 6-8 % to the second chain's launches (profiles/r05_sweep_split_chain.txt) -- taken out in round 6. */
    if (phase == 3) return;
    hipLaunchKernelGGL(k_bwt_fix0, g, dim3(LBZ_BWT_WG), 0, q, (const u8 *)c->T, c->B, c->meta, c->L,
                       first, count, nblk, segs, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst, 0u);
    for (u32 r = 0; r < R; r++)
      hipLaunchKernelGGL(k_bwt_fixr, g, dim3(LBZ_BWT_WG), 0, q, (const u8 *)c->T, c->B, c->meta, c->L,
                         first, count, nblk, segs, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst, r, 0u);
    hipLaunchKernelGGL(k_bwt_fixend, dim3(nblk), dim3(LBZ_BWT_WG), 0, q, c->B, c->meta, c->L,
                       first, count, ws, (u64)c->slot_bytes, wsp, (u64)c->spill_bytes, lst, 0u);
  }
}

/* Stream assembly of a chunk, round by round (compress.c:238-250, :291-321): as soon as a round's blocks are packed and the
 * round before it has its offsets, the round's blocks get theirs (k_offsets continues the stream position and the CRC fold
 * in lbz_stream_state) and are copied to their place in `out` (k_gather) -- on the round's own stream, beside the kernels of
 * the later rounds.  `out` may be page-locked HOST memory (the host-buffer calls): the stream then crosses PCIe while the
 * device still sorts, and nothing is left to copy at the end.                                                          */
struct finish_plan {
  u8 *out;
  u64 out_cap;
  bool first, last, body;       /* this chunk opens / closes the stream; blocks only (no header, no trailer) */
  bool host_out;                /* `out` is page-locked host memory */
};

static int run_chunk(lbzamd_ctx *c, const u8 *d_in, size_t len, uint32_t nsl, int upto, bool collected = false,
                     const finish_plan *fin = nullptr)
{
  hipStream_t s = c->stream;
  HIPCHK(hipEventRecord(c->ev[0], s));
  size_t nbev = 0;
  if (upto < 1) {
    if (timed_begin(c, &nbev, 5, s)) return -1;
    hipLaunchKernelGGL(k_collect, dim3(nsl), dim3(LBZ_COLLECT_WG), 0, s, d_in, (u64)len, c->L, c->T, c->meta, 0u, nullptr, nullptr);
    if (timed_end(c, &nbev, s)) return -1;
  }
  HIPCHK(hipEventRecord(c->ev[1], s));
  if (upto >= 1) {
    /* the rounds: slab ranges of at most nslots slabs, in slab order when the stream is assembled round by round (each
       round continues the previous one's stream position), of equal size; a host-buffer call begins with a short round,
       so that the device starts after 2 ms of PCIe traffic instead of waiting for a third of the input */
    std::vector<std::pair<uint32_t, uint32_t>> plan;
    const uint32_t head = LBZ_HEAD_SLABS;
    bool has_head = false;
    const bool host_in = c->h2d_host && !collected;
    if (fin) {
      uint32_t at = 0;
      if (host_in && nsl > 4u * head && nsl > c->nslots) {
        /* the head round's lane and workspaces (5 GB at -9, outside the budget that sized the slots): without them -- a
           small or busy device -- the call simply has no head round */
        bool ok = true;
        if (!c->head_q) ok = hipStreamCreate(&c->head_q) == hipSuccess;
        if (ok && !c->head_ev) ok = hipEventCreateWithFlags(&c->head_ev, hipEventDisableTiming) == hipSuccess;
        if (ok && !c->ws_head) ok = hipMalloc((void **)&c->ws_head, (size_t)head * (c->slot_bytes + c->spill_bytes)) == hipSuccess;
        if (ok) { plan.push_back({ 0u, head }); at = head; has_head = true; }
        else (void)hipGetLastError();
      }
      const uint32_t rem = nsl - at, nr = (rem + c->nslots - 1u) / c->nslots, per = nr ? (rem + nr - 1u) / nr : 0u;
      for (uint32_t k = 0; k < nr; k++) {
        const uint32_t f = at + k * per;
        plan.push_back({ f, nsl - f < per ? nsl - f : per });
      }
    } else {
      const uint32_t nr = (nsl + c->nslots - 1u) / c->nslots;
      const bool short_last = (nsl % c->nslots) != 0u;                       /* issue the short round first */
      for (uint32_t i = 0; i < nr; i++) {
        const uint32_t r = short_last ? (i == 0 ? nr - 1u : i - 1u) : i;
        const uint32_t f = r * c->nslots;
        plan.push_back({ f, nsl - f < c->nslots ? nsl - f : c->nslots });
      }
    }
    const uint32_t nrounds = (uint32_t)plan.size();
    const bool two = c->nstreams > 1 && nrounds > 1;
    if (two) for (unsigned k = 0; k + 1 < c->nstreams; k++) HIPCHK(hipStreamWaitEvent(c->side[k], c->ev[1], 0));
    if (has_head) HIPCHK(hipStreamWaitEvent(c->head_q, c->ev[1], 0));
    bool pinned_in = true;
    auto issue_copy = [&](uint32_t i) -> int {
      const size_t o = (size_t)plan[i].first * c->L.M;
      const size_t nb = (size_t)plan[i].second * c->L.M < len - o ? (size_t)plan[i].second * c->L.M : len - o;
      HIPCHK(hipMemcpyAsync(const_cast<u8 *>(d_in) + o, c->h2d_host + (d_in - c->d_in) + o, nb, hipMemcpyHostToDevice, c->copy_q));
      HIPCHK(hipEventRecord(c->cev[i], c->copy_q));
      return 0;
    };
    if (host_in) {
      if (!c->copy_q) HIPCHK(hipStreamCreate(&c->copy_q));
      while (c->cev.size() < nrounds) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->cev.push_back(e);
      }
      HIPCHK(hipStreamWaitEvent(c->copy_q, c->ev[1], 0));
      /* Page-locked input: every round's copy is issued now, in round order, on the one copy stream (copies on several
         streams share the link and all arrive late).  Pageable input (bytes, numpy arrays): hipMemcpyAsync stages such a
         copy and holds the calling thread until it is done, so a round's copy is issued next to the launches of the round
         before it -- the device works on round i while round i + 1 crosses the link.                                 */
      hipPointerAttribute_t pa;
      pinned_in = hipPointerGetAttributes(&pa, c->h2d_host) == hipSuccess && pa.type == hipMemoryTypeHost;
      if (!pinned_in) (void)hipGetLastError();
      for (uint32_t i = 0; i < (pinned_in ? nrounds : (nrounds ? 1u : 0u)); i++)
        if (issue_copy(i)) return -1;
    }
    while (fin && c->fev.size() < nrounds) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      c->fev.push_back(e);
    }
    for (uint32_t i = 0; i < nrounds; i++) {
      const uint32_t first = plan[i].first;
      const uint32_t count = plan[i].second;
      const uint32_t grid = 2u * count;
      const bool is_head = has_head && i == 0;
      const uint32_t li = has_head ? i - 1u : i;                  /* the head round does not take a lane */
      const unsigned lane = (two && !is_head) ? li % c->nstreams : 0u;
      hipStream_t q = is_head ? c->head_q : (lane ? c->side[lane - 1u] : s);
      u8 *ws = is_head ? c->ws_head : c->ws + (size_t)lane * c->nslots * (c->slot_bytes + c->spill_bytes);
      u8 *wsp = ws + (size_t)(is_head ? head : c->nslots) * c->slot_bytes;
      if (host_in) HIPCHK(hipStreamWaitEvent(q, c->cev[i], 0));   /* the round's slabs have arrived */
      if (!collected) {
        if (timed_begin(c, &nbev, 5, q)) return -1;
        hipLaunchKernelGGL(k_collect, dim3(count), dim3(LBZ_COLLECT_WG), 0, q, d_in, (u64)len, c->L, c->T, c->meta, first, nullptr, nullptr);
        if (timed_end(c, &nbev, q)) return -1;
      }
      if (timed_begin(c, &nbev, 0, q)) return -1;
      launch_sort(c, q, first, count, grid, ws, wsp, nullptr, 0, two);
      if (timed_end(c, &nbev, q) || timed_begin(c, &nbev, 1, q)) return -1;
      launch_sort(c, q, first, count, grid, ws, wsp, nullptr, 1);
      if (timed_end(c, &nbev, q) || timed_begin(c, &nbev, 2, q)) return -1;
      launch_sort(c, q, first, count, grid, ws, wsp, nullptr, 2);
      if (timed_end(c, &nbev, q)) return -1;
      if (upto >= 2) {
        if (timed_begin(c, &nbev, 3, q)) return -1;
        launch_mtf(c, q, first, count, grid, count, nullptr);
        if (timed_end(c, &nbev, q)) return -1;
      }
      if (upto >= 3) {
        if (timed_begin(c, &nbev, 4, q)) return -1;
        hipLaunchKernelGGL(k_encode, dim3(grid), dim3(LBZ_ENCODE_WG), 0, q, (const u16 *)c->V, (const u32 *)c->freq, c->O, c->meta, c->L, first, count, nullptr);
        if (timed_end(c, &nbev, q)) return -1;
      }
      if (fin && upto >= 3) {
        if (i > 0) HIPCHK(hipStreamWaitEvent(q, c->fev[i - 1u], 0));
        hipLaunchKernelGGL(k_offsets, dim3(1), dim3(LBZ_FINISH_WG), 0, q, (const lbz_block_meta *)(c->meta + 2u * (size_t)first), grid,
                           (u32)c->bs100k, (u32)(fin->first && i == 0), (u32)(fin->last && i + 1u == nrounds), (u32)fin->body,
                           c->offs + 2u * (size_t)first, c->st, fin->out, fin->out_cap);
        HIPCHK(hipEventRecord(c->fev[i], q));
        hipLaunchKernelGGL(k_gather, dim3(fin->host_out && grid > 64u ? 64u : grid), dim3(LBZ_FINISH_WG), 0, q,
                           (const u8 *)(c->O + lbz_out_off(c->L, 2u * first)),
                           (const lbz_block_meta *)(c->meta + 2u * (size_t)first), c->L, (const u64 *)(c->offs + 2u * (size_t)first),
                           (const lbz_stream_state *)c->st, fin->out, count);
      }
      if (host_in && !pinned_in && i + 1u < nrounds && issue_copy(i + 1u)) return -1;
    }
    if (has_head) {
      HIPCHK(hipEventRecord(c->head_ev, c->head_q));
      HIPCHK(hipStreamWaitEvent(s, c->head_ev, 0));
    }
    if (two)
      for (unsigned k = 0; k + 1 < c->nstreams; k++) {
        while (c->jev.size() <= k) {
          hipEvent_t e;
          HIPCHK(hipEventCreate(&e));
          c->jev.push_back(e);
        }
        HIPCHK(hipEventRecord(c->jev[k], c->side[k]));
        HIPCHK(hipStreamWaitEvent(s, c->jev[k], 0));
      }
  }
  c->nbev_used = nbev;
  HIPCHK(hipEventRecord(c->ev[2], s));
  HIPCHK(hipGetLastError());
  c->last_nslabs = nsl;
  return 0;
}

/* rounds (ev0..ev2, wall), finish (ev2..ev3); per-kernel sums from the pairs */
static int add_times(lbzamd_ctx *c, float *acc)
{
  for (int i = 0; i < 3; i++) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]));
    acc[i] += t;
  }
  for (size_t r = 0; r + 1 < c->nbev_used; r += 2) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, c->bev[r], c->bev[r + 1]));
    c->kms[c->bkind[r / 2]] += t;
  }
  return 0;
}

static int compress_device(lbzamd_ctx *c, const void *d_in_v, size_t len, void *d_out_v, size_t out_cap,
                           size_t *out_len, bool body, lbzamd_part *part);

/* -u / --sequential of the reference (main.c, compress.c:129-198): blocks take input until they are full instead
 * of being cut at every bs100k * 100000 input bytes -- the blocking of bzip2 itself, a slightly better ratio, and
 * a serial dependency between the blocks' starts. */
extern "C" int lbzamd_set_sequential(lbzamd_ctx *c, int on)
{
  if (!c) { g_err = "lbzamd_set_sequential: bad argument"; return -1; }
  c->sequential = on != 0;
  return 0;
}

extern "C" int lbzamd_compress_device(lbzamd_ctx *c, const void *d_in_v, size_t len,
                                      void *d_out_v, size_t out_cap, size_t *out_len)
{
  return compress_device(c, d_in_v, len, d_out_v, out_cap, out_len, false, nullptr);
}

/* The blocks of a slab range only -- no stream header, no trailer -- plus what the muxer of a
 * multi-GPU job needs to splice the ranges into ONE stream (compress.c:238-250, :291-321): the
 * number of blocks and the CRC fold of the range started from zero.  The fold cc' = rotl(cc,1) ^ ~c
 * (encode.h:38) is linear over GF(2): after a range of m blocks, cc = rotl(cc_before, m mod 32) ^ fold. */
extern "C" int lbzamd_compress_device_body(lbzamd_ctx *c, const void *d_in_v, size_t len,
                                           void *d_out_v, size_t out_cap, size_t *out_len, lbzamd_part *part)
{
  if (!part) { g_err = "lbzamd_compress_device_body: bad argument"; return -1; }
  return compress_device(c, d_in_v, len, d_out_v, out_cap, out_len, true, part);
}

static int compress_device(lbzamd_ctx *c, const void *d_in_v, size_t len, void *d_out_v, size_t out_cap,
                           size_t *out_len, bool body, lbzamd_part *part)
{
  if (!c || !d_out_v || !out_len || (len && !d_in_v)) { g_err = "lbzamd_compress_device: bad argument"; return -1; }
  HIPCHK(hipSetDevice(c->device));
  const u8 *d_in = (const u8 *)d_in_v;
  u8 *d_out = (u8 *)d_out_v;
  const uint32_t M = c->L.M;
  const size_t nslabs = (len + M - 1u) / M;
  float acc[6] = { 0, 0, 0, 0, 0, 0 };
  hipStream_t s = c->stream;
  for (float &k : c->kms) k = 0;
  c->stats.seq_fast_links = 0;

  if (c->sequential) {
    /* -u / --sequential (compress.c:129-198): chunks of up to max_slabs BLOCKS; where a block starts is known
       when its predecessor has been cut, so the tokenising pass of a chunk's blocks runs as a chain inside one
       launch (k_collect_seq) before the rounds take the blocks through the other stages */
    if (body) { g_err = "lbzamd_compress: a slab range of a sequential stream cannot be cut ahead of time (body-only call in sequential mode)"; return -1; }
    if (!c->seq_starts) {
      HIPCHK(hipMalloc((void **)&c->seq_starts, ((size_t)c->max_slabs + 2u) * sizeof(unsigned long long)));
      HIPCHK(hipMalloc((void **)&c->seq_ticket, sizeof(u32)));
      HIPCHK(hipMalloc((void **)&c->seq_out, sizeof(lbz_seq_out)));
    }
    if (c->h2d_host) {                           /* host-buffer call: the cuts are not known slab by slab, copy up front */
      HIPCHK(hipMemcpyAsync(const_cast<u8 *>(d_in), c->h2d_host, len, hipMemcpyHostToDevice, s));
    }
    /* what every 32 KB step of the input emits, and the prefix sums: a link of the block chain then needs two short
       passes instead of a walk over the whole block (k_collect.hip, k_seq_tiles / k_seq_prefix) */
    const u32 a0 = (u32)((uintptr_t)d_in & 15u);
    const size_t ntiles = len ? ((size_t)a0 + len + LBZ_SEQ_STEP - 1u) / LBZ_SEQ_STEP : 0u;
    const bool tables = ntiles > 0 && ntiles < 0x7FFFFFFFu && !getenv("LBZAMD_SEQ_NO_TABLES");
    if (tables) {
      if (ntiles > c->seq_tab_cap) {
        (void)hipFree(c->seq_tab32); (void)hipFree(c->seq_tab64); c->seq_tab32 = nullptr; c->seq_tab64 = nullptr; c->seq_tab_cap = 0;
        HIPCHK(hipMalloc((void **)&c->seq_tab32, 3u * ntiles * sizeof(u32)));
        HIPCHK(hipMalloc((void **)&c->seq_tab64, (2u * ntiles + 1u) * sizeof(unsigned long long)));
        c->seq_tab_cap = ntiles;
      }
      hipLaunchKernelGGL(k_seq_tiles, dim3((u32)ntiles), dim3(LBZ_COLLECT_WG), 0, s, d_in, (u64)len,
                         c->seq_tab32, c->seq_tab32 + ntiles, c->seq_tab32 + 2u * ntiles);
      hipLaunchKernelGGL(k_seq_prefix, dim3(1), dim3(LBZ_COLLECT_WG), 0, s, (const u32 *)c->seq_tab32, (const u32 *)(c->seq_tab32 + ntiles),
                         (const u32 *)(c->seq_tab32 + 2u * ntiles), (u32)ntiles, a0, c->seq_tab64, c->seq_tab64 + ntiles);
    }
    uint64_t pos = 0;
    bool first = true;
    uint32_t nfast = 0;
    do {
      const unsigned long long start0 = pos + 1ull;
      HIPCHK(hipMemsetAsync(c->seq_starts, 0, ((size_t)c->max_slabs + 2u) * sizeof(unsigned long long), s));
      HIPCHK(hipMemsetAsync(c->seq_ticket, 0, sizeof(u32), s));
      HIPCHK(hipMemsetAsync(c->seq_out, 0, sizeof(lbz_seq_out), s));
      HIPCHK(hipMemcpyAsync(c->seq_starts, &start0, sizeof start0, hipMemcpyHostToDevice, s));
      HIPCHK(hipEventRecord(c->ev[4], s));
      if (len) hipLaunchKernelGGL(k_collect_seq, dim3(c->max_slabs), dim3(LBZ_COLLECT_WG), 0, s, d_in, (u64)len, c->L, c->T, c->meta,
                                  (u32)c->max_slabs, c->seq_starts, c->seq_ticket, c->seq_out, 0u,
                                  (const unsigned long long *)(tables ? c->seq_tab64 : nullptr),
                                  (const unsigned long long *)(tables ? c->seq_tab64 + ntiles : nullptr), (u32)ntiles);
      HIPCHK(hipEventRecord(c->ev[5], s));
      lbz_seq_out so{};
      if (len) {
        HIPCHK(hipMemcpyAsync(&so, c->seq_out, sizeof so, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipGetLastError());
        if (so.err) { g_err = "lbzamd_compress: the block chain of the sequential mode broke (device error)"; return -1; }
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, c->ev[4], c->ev[5]));
        acc[3] += t;
      }
      const uint32_t nb = so.nblocks;
      nfast += so.nfast;
      const bool last = !len || so.next >= len;
      if (nb) {
        if (run_chunk(c, d_in, len, nb, 3, true)) return -1;
      } else {
        c->nbev_used = 0;
        for (int i = 0; i <= 2; i++) HIPCHK(hipEventRecord(c->ev[i], s));
      }
      hipLaunchKernelGGL(k_offsets, dim3(1), dim3(LBZ_FINISH_WG), 0, s, (const lbz_block_meta *)c->meta, (u32)(2u * nb),
                         (u32)c->bs100k, (u32)first, (u32)last, 0u, c->offs, c->st, d_out, (u64)out_cap);
      if (nb)
        hipLaunchKernelGGL(k_gather, dim3((u32)(2u * nb)), dim3(LBZ_FINISH_WG), 0, s, (const u8 *)c->O,
                           (const lbz_block_meta *)c->meta, c->L, (const u64 *)c->offs,
                           (const lbz_stream_state *)c->st, d_out, (u32)nb);
      HIPCHK(hipEventRecord(c->ev[3], s));
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(s));
      if (add_times(c, acc)) return -1;
      if (!last && nb == 0) { g_err = "lbzamd_compress: sequential mode made no progress"; return -1; }
      pos = so.next;
      first = false;
      if (last) break;
    } while (true);
    c->kms[5] += acc[3];
    c->stats.seq_fast_links = nfast;
  }
  size_t done = 0;
  bool first = true;
  if (!c->sequential) do {
    const size_t nsl = nslabs - done < c->max_slabs ? nslabs - done : c->max_slabs;
    const bool last = done + nsl == nslabs;
    const size_t off = done * (size_t)M;
    const size_t clen = last ? len - off : nsl * (size_t)M;
    if (nsl) {
      const finish_plan fin = { d_out, (u64)out_cap, first, last, body, c->out_is_host };
      if (run_chunk(c, d_in + off, clen, (uint32_t)nsl, 3, false, &fin)) return -1;      /* assembles the stream round by round */
    } else {
      c->nbev_used = 0;
      for (int i = 0; i <= 2; i++) HIPCHK(hipEventRecord(c->ev[i], s));
      hipLaunchKernelGGL(k_offsets, dim3(1), dim3(LBZ_FINISH_WG), 0, s, (const lbz_block_meta *)c->meta, 0u,
                         (u32)c->bs100k, (u32)first, (u32)last, (u32)body, c->offs, c->st, d_out, (u64)out_cap);   /* the empty stream */
    }
    HIPCHK(hipEventRecord(c->ev[3], s));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    if (add_times(c, acc)) return -1;
    done += nsl;
    first = false;
  } while (done < nslabs);

  lbz_stream_state st;
  HIPCHK(hipMemcpy(&st, c->st, sizeof st, hipMemcpyDeviceToHost));
  c->stats.n_in = len; c->stats.n_rle = st.n_rle; c->stats.n_mtf = st.n_mtf; c->stats.n_out = st.pos;
  c->stats.sort_elems = st.sort_elems; c->stats.nblocks = st.nblocks; c->stats.nperiodic = st.nperiodic;
  /* per-kernel figures are sums over launches (launches of the two streams overlap, so they add
     up to more than the wall time); ms_total is wall time on the caller's stream */
  c->stats.ms_collect = c->kms[5];
  c->stats.ms_bwt_part = c->kms[0]; c->stats.ms_bwt_batch = c->kms[1]; c->stats.ms_bwt_fix = c->kms[2];
  c->stats.ms_bwt = c->kms[0] + c->kms[1] + c->kms[2];
  c->stats.ms_mtf = c->kms[3]; c->stats.ms_encode = c->kms[4]; c->stats.ms_finish = acc[2];
  c->stats.ms_total = acc[0] + acc[1] + acc[2] + acc[3];
  if (st.err) {
    char buf[96];
    snprintf(buf, sizeof buf, "device pipeline error code %u%s", st.err, st.err == 100u ? " (output buffer too small)" : "");
    g_err = buf;
    return -2;
  }
  *out_len = (size_t)st.pos;
  if (part) { part->nblocks = st.nblocks; part->crc_fold = st.crc; part->bytes = st.pos; }
  return 0;
}

extern "C" uint32_t lbzamd_fold_parts(uint32_t cc, const lbzamd_part *parts, size_t nparts)
{
  for (size_t i = 0; i < nparts; i++) {
    const uint32_t r = parts[i].nblocks & 31u;
    cc = (r ? ((cc << r) | (cc >> (32u - r))) : cc) ^ parts[i].crc_fold;
  }
  return cc;
}

static int ensure_staging(lbzamd_ctx *c, size_t in_bytes, size_t out_bytes)
{
  if (in_bytes > c->d_in_cap) {
    (void)hipFree(c->d_in); c->d_in = nullptr; c->d_in_cap = 0;
    HIPCHK(hipMalloc((void **)&c->d_in, in_bytes + 256));
    c->d_in_cap = in_bytes;
  }
  if (out_bytes > c->d_out_cap) {
    (void)hipFree(c->d_out); c->d_out = nullptr; c->d_out_cap = 0;
    HIPCHK(hipMalloc((void **)&c->d_out, out_bytes + 256));
    c->d_out_cap = out_bytes;
  }
  return 0;
}

static int compress_host(lbzamd_ctx *c, const uint8_t *in, size_t len, uint8_t *out, size_t out_cap, size_t *out_len,
                         bool body, lbzamd_part *part);
extern "C" int lbzamd_compress_host(lbzamd_ctx *c, const uint8_t *in, size_t len,
                                    uint8_t *out, size_t out_cap, size_t *out_len)
{
  return compress_host(c, in, len, out, out_cap, out_len, false, nullptr);
}
extern "C" int lbzamd_compress_host_body(lbzamd_ctx *c, const uint8_t *in, size_t len,
                                         uint8_t *out, size_t out_cap, size_t *out_len, lbzamd_part *part)
{
  if (!part) { g_err = "lbzamd_compress_host_body: bad argument"; return -1; }
  return compress_host(c, in, len, out, out_cap, out_len, true, part);
}
static int compress_host(lbzamd_ctx *c, const uint8_t *in, size_t len, uint8_t *out, size_t out_cap, size_t *out_len,
                         bool body, lbzamd_part *part)
{
  if (!c || !out || !out_len || (len && !in)) { g_err = "lbzamd_compress_host: bad argument"; return -1; }
  HIPCHK(hipSetDevice(c->device));
  const size_t bound = lbzamd_bound(len);
  /* a page-locked output buffer (lbzamd_pinned_alloc, hipHostMalloc, torch's pin_memory) is written by the device itself,
     round by round; anything else gets the stream through a device staging buffer and one copy at the end */
  void *mapped = nullptr;
  bool direct = !c->sequential && out_cap > 0 && !getenv("LBZAMD_NO_DIRECT_OUT")
                && hipHostGetDevicePointer(&mapped, out, 0) == hipSuccess && mapped != nullptr;
  if (!direct) (void)hipGetLastError();
  if (ensure_staging(c, len ? len : 1, direct ? 0 : bound)) return -1;
  size_t n = 0;
  c->h2d_host = in;                           /* run_chunk copies round by round */
  c->out_is_host = direct;
  const int rc = direct ? compress_device(c, c->d_in, len, mapped, out_cap, &n, body, part)
                        : compress_device(c, c->d_in, len, c->d_out, bound, &n, body, part);
  c->h2d_host = nullptr;
  c->out_is_host = false;
  if (rc) return rc;
  if (n > out_cap) { g_err = "lbzamd_compress_host: output buffer too small"; return -2; }
  if (!direct) HIPCHK(hipMemcpy(out, c->d_out, n, hipMemcpyDeviceToHost));
  *out_len = n;
  return 0;
}

/* page-locked host memory for the splitter/muxer of a host program (DMA at link rate, async copies) */
extern "C" void *lbzamd_pinned_alloc(size_t bytes)
{
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) return nullptr;   /* every device may DMA it */
  return p;
}
extern "C" int lbzamd_device_count(void)
{
  return logical_devices();
}
extern "C" void lbzamd_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int lbzamd_get_stats(lbzamd_ctx *c, lbzamd_stats *st)
{
  if (!c || !st) return -1;
  *st = c->stats;
  return 0;
}

extern "C" int lbzamd_run_stages(lbzamd_ctx *c, const uint8_t *in, size_t len, int upto)
{
  if (!c || !in || !len) { g_err = "lbzamd_run_stages: bad argument"; return -1; }
  HIPCHK(hipSetDevice(c->device));
  const uint32_t M = c->L.M;
  const size_t nsl = (len + M - 1u) / M;
  if (nsl > c->max_slabs) { g_err = "lbzamd_run_stages: input exceeds one chunk"; return -1; }
  if (ensure_staging(c, len, 0)) return -1;
  HIPCHK(hipMemcpyAsync(c->d_in, in, len, hipMemcpyHostToDevice, c->stream));
  if (run_chunk(c, c->d_in, len, (uint32_t)nsl, upto)) return -1;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" uint32_t lbzamd_block_slots(lbzamd_ctx *c) { return c ? 2u * c->last_nslabs : 0u; }

extern "C" int lbzamd_block_info_get(lbzamd_ctx *c, uint32_t blk, lbzamd_block_info *info)
{
  if (!c || !info || blk >= 2u * c->last_nslabs) { g_err = "lbzamd_block_info_get: bad argument"; return -1; }
  HIPCHK(hipSetDevice(c->device));
  lbz_block_meta m;
  HIPCHK(hipMemcpy(&m, c->meta + blk, sizeof m, hipMemcpyDeviceToHost));
  info->n = m.n; info->crc = m.crc; info->consumed = m.consumed; info->bwt_idx = m.bwt_idx;
  info->periodic = m.periodic; info->nmtf = m.nmtf; info->alpha = m.alpha; info->num_trees = m.num_trees;
  info->num_sel = m.num_sel; info->out_len = m.out_len; info->err = m.err; info->rounds = m.rounds;
  info->sort_elems = m.sort_elems; for (int i = 0; i < 8; i++) info->ticks[i] = m.ticks[i];
  for (int i = 0; i < 16; i++) info->fticks[i] = m.fticks[i];
  if (getenv("LBZAMD_DIAG_DEEP")) {       /* (tuning) the text rounds' counts in place of the rank rounds' ticks: tests/tools/diag_rows.py */
    for (unsigned i = 0; i <= LBZ_DEEP_ROUNDS && i < 9u; i++) info->fticks[i] = m.deep_tot[i];
    info->fticks[9] = m.deep_h0; info->fticks[10] = m.deep_skip; info->fticks[11] = m.deep_long; info->fticks[12] = m.deep_rows;
    info->fticks[13] = m.deep_hmin[1]; info->fticks[14] = m.deep_hmin[LBZ_DEEP_ROUNDS];
  }
  memcpy(info->inuse, m.inuse, 256);
  return 0;
}

extern "C" long lbzamd_read_stage(lbzamd_ctx *c, uint32_t blk, int stage, void *dst, size_t cap)
{
  if (!c || !dst || blk >= 2u * c->last_nslabs) { g_err = "lbzamd_read_stage: bad argument"; return -1; }
  if (hipSetDevice(c->device) != hipSuccess) return -1;
  lbz_block_meta m;
  if (hipMemcpy(&m, c->meta + blk, sizeof m, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  const size_t eo = lbz_elem_off(c->L, blk);
  const void *src = nullptr;
  size_t bytes = 0;
  switch (stage) {
    case LBZAMD_STAGE_RLE:  src = c->T + eo; bytes = m.n; break;
    case LBZAMD_STAGE_BWT:  src = c->B + eo; bytes = m.n; break;
    case LBZAMD_STAGE_MTFV: src = c->V + eo; bytes = 2u * (size_t)m.nmtf; break;
    case LBZAMD_STAGE_OUT:  src = c->O + lbz_out_off(c->L, blk); bytes = m.out_len; break;
    default: g_err = "lbzamd_read_stage: bad stage"; return -1;
  }
  if (bytes > cap) { g_err = "lbzamd_read_stage: buffer too small"; return -1; }
  if (bytes && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long)bytes;
}

/* ===================================================================== the inverse path (C) */
#include <algorithm>

struct lbzamd_dctx {
  int device = 0;
  uint32_t max_blocks = 0, cap = 0, ncus = 0; /* cap: elements per block in the per-block arrays */
  hipStream_t q = nullptr;
  hipEvent_t ev[7] = {};
  u8 *tt8 = nullptr, *W = nullptr, *X = nullptr;   /* X: the second half of what the walk's first pass keeps of every sublist */
  u32 *tt = nullptr, *nmarks = nullptr, *pinfo = nullptr;
  u64 *marks = nullptr;
  lbz_dblock *blocks = nullptr;
  u8 *d_in = nullptr, *d_out = nullptr;
  size_t d_in_cap = 0, d_out_cap = 0;
  uint32_t marks_cap = 0;
  lbzamd_dresume *rs = nullptr;              /* lbzamd_decompress_window: this call takes one window of a longer input (state in, state out) */
  bool rs_final = true;
  bool grow_out = false;                     /* lbzamd_decompress_alloc: the output buffer (d_out) grows with what the blocks turn out to hold */
  u8 *h_in = nullptr, *h_out = nullptr;      /* the work-unit interface: page-locked staging of one block's bits and bytes (copies of pageable
                                                memory wait for the whole device, i.e. for every other worker thread's block) */
  size_t h_in_cap = 0, h_out_cap = 0;
  lbzamd_dstats stats{};
};

extern "C" void lbzamd_ddestroy(lbzamd_dctx *c)
{
  if (!c) return;
  (void)hipFree(c->tt8); (void)hipFree(c->W); (void)hipFree(c->X); (void)hipFree(c->tt);
  (void)hipFree(c->pinfo); (void)hipFree(c->nmarks); (void)hipFree(c->marks); (void)hipFree(c->blocks); (void)hipFree(c->d_in); (void)hipFree(c->d_out);
  for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->q) (void)hipStreamDestroy(c->q);
  if (c->h_in) (void)hipHostFree(c->h_in);
  if (c->h_out) (void)hipHostFree(c->h_out);
  delete c;
}

extern "C" int lbzamd_dcreate(lbzamd_dctx **out, int device, unsigned max_blocks)
{
  if (!out || max_blocks < 1) { g_err = "lbzamd_dcreate: bad argument"; return -1; }
  const int ndev = logical_devices();
  if (ndev < 1) { g_err = "lbzamd_dcreate: no HIP device (this library has no CPU path)"; return -1; }
  if (device < 0) HIPCHK(hipGetDevice(&device));
  if (device >= ndev) { g_err = "lbzamd_dcreate: no such device"; return -1; }
  device = physical_of(device);
  HIPCHK(hipSetDevice(device));
  lbzamd_dctx *c = new lbzamd_dctx;
  c->device = device;
  { hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device)); c->ncus = (uint32_t)prop.multiProcessorCount; }
  c->max_blocks = max_blocks;
  c->cap = round_up(LBZ_MAX_BLOCK + 64u, 256u);
  c->marks_cap = 1u << 20;
#define DALLOC(p, bytes) do { hipError_t e_ = hipMalloc((void **)&(p), (bytes)); \
    if (e_ != hipSuccess) { lbzamd_ddestroy(c); return fail_msg("hipMalloc " #p, e_); } } while (0)
  DALLOC(c->tt8, (size_t)max_blocks * c->cap);
  DALLOC(c->W, (size_t)max_blocks * c->cap);
  DALLOC(c->X, (size_t)max_blocks * c->cap);
  DALLOC(c->tt, (size_t)max_blocks * c->cap * sizeof(u32));
  DALLOC(c->pinfo, (size_t)max_blocks * (c->cap / 16u) * sizeof(u32));
  DALLOC(c->blocks, (size_t)max_blocks * sizeof(lbz_dblock));
  DALLOC(c->marks, (size_t)c->marks_cap * sizeof(u64));
  DALLOC(c->nmarks, sizeof(u32));
#undef DALLOC
  hipError_t e = hipStreamCreate(&c->q);
  if (e != hipSuccess) { lbzamd_ddestroy(c); return fail_msg("hipStreamCreate", e); }
  for (auto &ev : c->ev) { e = hipEventCreate(&ev); if (e != hipSuccess) { lbzamd_ddestroy(c); return fail_msg("hipEventCreate", e); } }
  *out = c;
  return 0;
}

static uint32_t rd_be32_bits(const std::vector<uint8_t> &h, uint64_t bit)
{
  /* 32 bits at an arbitrary bit position of a host copy (the stream trailers) */
  uint64_t v = 0;
  const uint64_t by = bit >> 3;
  for (int i = 0; i < 5; i++) v = (v << 8) | (by + i < h.size() ? h[by + i] : 0u);
  return (uint32_t)(v >> (8u - (bit & 7u)));
}

extern "C" int lbzamd_decompress_device(lbzamd_dctx *c, const void *d_in_v, size_t len, void *d_out_v, size_t out_cap, size_t *out_len)
{
  if (!c || !out_len || (len && !d_in_v)) { g_err = "lbzamd_decompress_device: bad argument"; return -1; }
  HIPCHK(hipSetDevice(c->device));
  const u8 *d_in = (const u8 *)d_in_v;
  u8 *d_out = (u8 *)d_out_v;
  hipStream_t q = c->q;
  c->stats = lbzamd_dstats{};
  c->stats.n_in = len;
  g_err_code = 0;
  *out_len = 0;
  lbzamd_dresume *const rs = c->rs;
  const bool windowed = rs != nullptr && !(rs->started == 0u && c->rs_final);     /* (a first window that is the last one too: the whole input) */
  const bool final = !windowed || c->rs_final;
  const uint64_t lenbits = (uint64_t)len * 8u;
  /* the reference takes its input in 32-bit words, the last one filled up with zero bytes (expand.c:835-842): where that
     filling ends, in this buffer's bits (a later window begins rs->base_bytes into the file) */
  const uint64_t base_bytes = windowed ? rs->base_bytes : 0u;
  const uint64_t padbits = ((base_bytes + (uint64_t)len + 3u) / 4u * 4u - base_bytes) * 8u;
  if (windowed && !final && len < 14) { rs->consumed_bit = rs->started && rs->in_stream ? rs->consumed_bit & 7u : 0u; return 0; }      /* nothing whole in here yet */
  if (len < 14 && !(windowed && rs->started)) {
    /* no room for a header and a trailer.  As the reference tells the two apart (process.c:664-681, expand.c:435): without
       "BZh1".."BZh9" in front it is not a bzip2 file, with it the file ends too early */
    uint8_t h4[16] = { 0 };
    if (len) HIPCHK(hipMemcpy(h4, d_in, len, hipMemcpyDeviceToHost));
    const bool hdr = len >= 4 && h4[0] == 'B' && h4[1] == 'Z' && h4[2] == 'h' && h4[3] >= '1' && h4[3] <= '9';
    g_err = "lbzamd_decompress: not a bzip2 stream (too short)";
    g_err_code = RE_MAGIC;
    if (hdr) {
      /* what follows the header, a 16-bit word at a time over the input filled up to 32-bit words (parse.c:152-262,
         expand.c:835-842): the first word that does not fit is a bad magic, the first that is not there the end of the file */
      const size_t words = (len + 3u) / 4u * 2u;                         /* 16-bit words there are, the zero filling included */
      static const unsigned blk[3] = { 0x3141, 0x5926, 0x5359 }, eos[3] = { 0x1772, 0x4538, 0x5090 };
      auto word = [&](size_t k) -> int { return k < words ? (int)((unsigned)h4[2 * k] << 8 | h4[2 * k + 1]) : -1; };
      g_err_code = RE_EOF;
      const unsigned *m = word(2) == (int)eos[0] ? eos : blk;
      for (size_t k = 0; k < 3u && g_err_code == RE_EOF; k++) {
        if (word(2 + k) < 0) break;
        if (word(2 + k) != (int)m[k]) g_err_code = RE_HEADER;
      }
      /* 13 bytes are 16 with the filling: the CRC words are "there".  A block header with nothing behind it ends at what is
         not a magic; an end-of-stream marker with a CRC other than zero is a wrong CRC (with three zero bytes the reference
         misses the fourth: the end of the file) */
      if (g_err_code == RE_EOF && words >= 8u && word(4) >= 0) {
        if (m == blk) g_err_code = RE_HEADER;
        else if (h4[10] | h4[11] | h4[12]) g_err_code = RE_STRMCRC;
      }
    }
    return -3;
  }
  /* 1. magics */
  HIPCHK(hipEventRecord(c->ev[0], q));
  HIPCHK(hipMemsetAsync(c->nmarks, 0, sizeof(u32), q));
  hipLaunchKernelGGL(k_dscan, dim3(LBZ_DSCAN_GRID((u64)len)), dim3(256), 0, q, d_in, (u64)len, c->marks, c->nmarks, c->marks_cap);
  HIPCHK(hipEventRecord(c->ev[1], q));
  u32 nm = 0;
  HIPCHK(hipMemcpyAsync(&nm, c->nmarks, sizeof nm, hipMemcpyDeviceToHost, q));
  HIPCHK(hipStreamSynchronize(q));
  if (nm > c->marks_cap) {                      /* more magics than the list holds (many tiny blocks): grow it and scan again */
    (void)hipFree(c->marks); c->marks = nullptr;
    HIPCHK(hipMalloc((void **)&c->marks, (size_t)nm * sizeof(u64)));
    c->marks_cap = nm;
    HIPCHK(hipMemsetAsync(c->nmarks, 0, sizeof(u32), q));
    hipLaunchKernelGGL(k_dscan, dim3(LBZ_DSCAN_GRID((u64)len)), dim3(256), 0, q, d_in, (u64)len, c->marks, c->nmarks, c->marks_cap);
    HIPCHK(hipMemcpyAsync(&nm, c->nmarks, sizeof nm, hipMemcpyDeviceToHost, q));
    HIPCHK(hipStreamSynchronize(q));
    if (nm > c->marks_cap) { g_err = "lbzamd_decompress: the list of magics changed between two scans"; return -1; }
  }
  std::vector<u64> marks(nm);
  if (nm) HIPCHK(hipMemcpy(marks.data(), c->marks, nm * sizeof(u64), hipMemcpyDeviceToHost));
  std::sort(marks.begin(), marks.end());
  /* 2. candidates.  A 48-bit magic can also occur by chance (or by design) inside a block's payload, so a mark is
     only a HINT, as parse.c's scan() is for the reference: every block mark is decoded as a candidate, and what the
     stream really consists of is decided afterwards by walking the chain header -> block -> (where that block's last
     code ended) next magic -> ... -> end-of-stream magic (step 4).  Here the candidates only need the block size
     limit of the stream they would belong to: the level of the last plausible stream header before them.        */
  std::vector<uint8_t> head(4);
  std::vector<lbz_dblock> hb;
  std::vector<long> cand_of(marks.size(), -1);
  auto header_at = [&](uint64_t sbyte, unsigned *level) -> int {       /* 1 = "BZh[1-9]" at sbyte, 0 = something else, -1 = HIP error */
    if (sbyte + 4 > len) return 0;
    if (hipMemcpy(head.data(), d_in + sbyte, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (head[0] != 'B' || head[1] != 'Z' || head[2] != 'h' || head[3] < '1' || head[3] > '9') return 0;
    *level = (unsigned)(head[3] - '0');
    return 1;
  };
  /* What the reference's parser says when the chain ends at bit `at` with no magic there (parse.c:152-262): it takes a header
     16 bits at a time and stops at the first word that does not fit -- ERR_HEADER -- or that is not all there -- ERR_EOF. */
  auto no_magic_at = [&](uint64_t at) -> int {
    const uint64_t nbits = padbits, by = at >> 3;
    std::vector<uint8_t> h(12, 0);
    if (by < len && hipMemcpy(h.data(), d_in + by, std::min<size_t>(11, len - by), hipMemcpyDeviceToHost) != hipSuccess) return RE_HEADER;
    auto word = [&](unsigned k) -> int { return at + 16u * (k + 1u) > nbits ? -1 : (int)(rd_be32_bits(h, (at & 7u) + 16u * k) >> 16); };
    static const int blk[3] = { 0x3141, 0x5926, 0x5359 }, eos[3] = { 0x1772, 0x4538, 0x5090 };
    const int w0 = word(0);
    if (w0 < 0) return RE_EOF;
    const int *m = w0 == eos[0] ? eos : blk;
    for (unsigned k = 0; k < 3u; k++) {
      const int w = word(k);
      if (w < 0) return RE_EOF;
      if (w != m[k]) return RE_HEADER;
    }
    return RE_EOF;                               /* a whole magic and no mark: its CRC words are cut off */
  };
  unsigned level0 = 0;
  bool resume_finished = false;                 /* a later window that begins behind a closed stream with something that is no header */
  {
    int h = 1;
    if (windowed && rs->started && rs->in_stream) level0 = rs->level;            /* a later window, in the middle of a stream */
    else h = header_at(0, &level0);
    if (h < 0) return fail_msg("hipMemcpy", hipGetLastError());
    if (h == 0 && windowed && rs->started) { resume_finished = true; h = 1; level0 = 9; }     /* trailing garbage is ignored, as bzip2 does */
    if (h == 0) { g_err = "lbzamd_decompress: not a bzip2 stream (bad header)"; g_err_code = RE_MAGIC; return -3; }
    unsigned lvl = level0;
    for (size_t i = 0; i < marks.size(); i++) {
      const uint64_t bit = marks[i] >> 1;
      if (bit < 32 && !(windowed && rs->started && rs->in_stream)) continue;      /* (inside the stream header -- unless this window begins in the middle of a stream) */
      if (marks[i] & 1u) {
        unsigned l2;
        const int h2 = header_at((bit + 48 + 32 + 7) / 8, &l2);
        if (h2 < 0) return fail_msg("hipMemcpy", hipGetLastError());
        if (h2 == 1) lvl = l2;
        continue;
      }
      lbz_dblock b{};
      b.bit_start = bit + 48;
      b.max_block = 900000u;                /* the most a block may hold: which stream a candidate belongs to (a mark inside a payload
                                               may have passed for a stream's end in front of it) is known only on the chain, where
                                               the block's length is held against its stream's size */
      (void)lvl;
      cand_of[i] = (long)hb.size();
      hb.push_back(b);
    }
  }
  /* 3. + 4. candidates decoded max_blocks at a time; behind every batch the chain is walked as far as it is decoded */
  struct trailer { uint64_t bit; uint32_t cc; };
  std::vector<trailer> trailers;
  float ms[6] = { 0, 0, 0, 0, 0, 0 };
  { float t = 0; HIPCHK(hipEventElapsedTime(&t, c->ev[0], c->ev[1])); ms[0] = t; }
  uint64_t total = 0;
  size_t mi = 0;                                /* next mark the walk looks at */
  uint64_t expect = 32;                         /* bit position at which the chain's next magic must sit */
  unsigned level = level0;
  bool in_stream = true, finished = false;      /* finished: the last stream is closed and what follows is not a header */
  uint32_t cc = 0, nblocks = 0, stream_blocks = 0;
  if (windowed && rs->started && rs->in_stream) { expect = rs->consumed_bit & 7u; cc = rs->cc; stream_blocks = rs->stream_blocks; }
  if (resume_finished || (windowed && rs->finished)) { finished = true; in_stream = false; }
  /* windows: a block (or an end-of-stream marker) is taken only if what must follow it can be seen -- the next magic, the
     marker's CRC, the next stream's header; else the window's walk ends in front of it and the next window begins there.
     What lies more than CUT_BITS in front of the window's end is not cut off by it: it is judged as the whole input's is. */
  const uint64_t CUT_BITS = 24u << 20;          /* 3 MB: no block is longer */
  bool carry = false;                           /* the walk ended at a cut: no error, `expect` is where the next window begins */
  auto near_end = [&](uint64_t bit) { return windowed && !final && bit + CUT_BITS > lenbits; };
  auto mark_at = [&](uint64_t bit) {             /* is a magic of either kind known at `bit`, all 48 bits of it in the window */
    auto it = std::lower_bound(marks.begin(), marks.end(), bit << 1);
    return it != marks.end() && (*it >> 1) == bit;
  };
  int pend_code = 0;                            /* the first block-level error met on the chain (see the walk) */
  std::string pend_msg;
  /* On an error (-3) the bytes IN FRONT of it are still delivered, as the reference has written what it decoded by then:
     *out_len bytes, the whole blocks in front of the first one that is refused.  `taken`: the candidates whose bytes count. */
  std::vector<uint8_t> taken(hb.size(), 0);
  uint64_t pend_total = 0, emitted = 0;
  bool stop = false;                            /* the walk met an error of the parser's kind: the chain ends there */
  for (size_t b0 = 0; b0 < hb.size() || b0 == 0; b0 += c->max_blocks) {
    const u32 nb = (u32)std::min<size_t>(c->max_blocks, hb.size() - b0);
    if (nb) {
      HIPCHK(hipMemcpyAsync(c->blocks, hb.data() + b0, nb * sizeof(lbz_dblock), hipMemcpyHostToDevice, q));
      HIPCHK(hipEventRecord(c->ev[1], q));
      /* a block per CU at most: 1024 threads each (one such workgroup is what a CU holds: 85 VGPRs); up to two per CU: 512
         threads; more: 256.  LBZAMD_DWIDE=0/1/2 forces 256 / 1024 / 512 */
      const char *dw = getenv("LBZAMD_DWIDE");
      const int width = dw ? dw[0] - '0' : (nb <= c->ncus ? 1 : (nb <= 2u * c->ncus ? 2 : 0));
      if (width == 1) hipLaunchKernelGGL(k_dblock_w, dim3(nb), dim3(1024), 0, q, d_in, (u64)len, c->blocks, nb, c->tt8, c->tt, c->W, c->pinfo, c->X, c->cap);
      else if (width == 2) hipLaunchKernelGGL(k_dblock_m, dim3(nb), dim3(512), 0, q, d_in, (u64)len, c->blocks, nb, c->tt8, c->tt, c->W, c->pinfo, c->X, c->cap);
      else hipLaunchKernelGGL(k_dblock, dim3(nb), dim3(256), 0, q, d_in, (u64)len, c->blocks, nb, c->tt8, c->tt, c->W, c->pinfo, c->X, c->cap);
      HIPCHK(hipEventRecord(c->ev[2], q));
      HIPCHK(hipMemcpyAsync(hb.data() + b0, c->blocks, nb * sizeof(lbz_dblock), hipMemcpyDeviceToHost, q));
      HIPCHK(hipStreamSynchronize(q));
      HIPCHK(hipGetLastError());
      /* the three stages of a block run back to back in one kernel: the stage figures are the slowest
         block's (100 MHz ticks), the pass as a whole is timed by events */
      float t = 0;
      HIPCHK(hipEventElapsedTime(&t, c->ev[1], c->ev[2])); ms[5] += t;
      u32 tk[6] = { 0, 0, 0, 0, 0, 0 };
      for (u32 i = 0; i < nb; i++) for (int k = 0; k < 6; k++) tk[k] = std::max(tk[k], hb[b0 + i].tk[k]);
      for (int k = 0; k < 3; k++) ms[1 + k] = std::max(ms[1 + k], tk[k] * 1e-5f);
      if (getenv("LBZAMD_DTIMES")) {
        double mhz = 0;
        for (u32 i = 0; i < nb; i++) if (hb[b0 + i].tk[3] == tk[3] && tk[3]) mhz = hb[b0 + i].cyc / (tk[3] * 1e-5 * 1e3);
        u32 wk[4] = { 0, 0, 0, 0 };
        for (u32 i = 0; i < nb; i++) for (int k = 0; k < 4; k++) wk[k] = std::max(wk[k], hb[b0 + i].wk[k]);
        fprintf(stderr, "lbzamd: walk parts, slowest: lengths %.2f ms, ranking %.2f, bytes %.2f, maps + CRC %.2f\n", wk[0] * 1e-5, wk[1] * 1e-5, wk[2] * 1e-5, wk[3] * 1e-5);
        fprintf(stderr, "lbzamd: k_dblock %u blocks, slowest: codes %.2f ms (bit chain %.2f at %.0f MHz, move-to-front chunks %.2f, scan + expansion %.2f), sort %.2f, walk %.2f\n",
                nb, tk[0] * 1e-5, tk[3] * 1e-5, mhz, tk[4] * 1e-5, tk[5] * 1e-5, tk[1] * 1e-5, tk[2] * 1e-5);
      }
    }
    /* the walk: marks in stream order, up to the last candidate of this batch */
    const bool last_batch = b0 + nb >= hb.size();
    for (; mi < marks.size(); mi++) {
      const long ci = cand_of[mi];
      if (ci >= (long)(b0 + nb)) break;                     /* not decoded yet */
      const uint64_t bit = marks[mi] >> 1;
      const bool is_end = (marks[mi] & 1u) != 0;
      if (finished || !in_stream || bit < expect) {         /* off the chain: inside a payload, a header or a trailer; or in trailing garbage */
        if (ci >= 0) { hb[ci].err = 99; hb[ci].out_len = 0; }
        continue;
      }
      if (bit > expect) {
        g_err = stream_blocks ? "lbzamd_decompress: no block or end-of-stream magic where the previous block ends (damaged or overrun block)"
                              : "lbzamd_decompress: no block magic behind the stream header";
        g_err_code = RE_HEADER;
        stop = true;
        break;
      }
      if (is_end) {
        const uint64_t sbyte = (bit + 48 + 32 + 7) / 8;
        if (windowed && !final && sbyte + 14 > len) { carry = true; break; }       /* its CRC, or what follows it, is not all here yet */
        trailers.push_back({ bit, cc });
        in_stream = false;
        unsigned l2;
        const int h2 = sbyte + 14 <= len ? header_at(sbyte, &l2) : 0;
        if (h2 < 0) return fail_msg("hipMemcpy", hipGetLastError());
        if (h2 == 1) { in_stream = true; level = l2; expect = (sbyte + 4) * 8; cc = 0; stream_blocks = 0; }
        else finished = true;                                  /* trailing garbage is ignored, as bzip2 does */
        continue;
      }
      lbz_dblock &b = hb[ci];
      if (near_end(bit) && !(b.bit_used + 48u <= lenbits && mark_at(b.bit_used))) { carry = true; break; }   /* perhaps cut off: the next window's */
      const bool kernel_ok = !b.err || b.err == 11u || b.err == 12u;   /* decoded to its last code (11: only the CRC differs; 12: it ends where a run's count should stand) */
      bool behind_the_block = b.err == 11u || b.err == 12u || b.err == 9u;   /* errors found with the block's bits all taken: the chain itself goes on */
      /* more bytes than the stream's block size allows: the muxer looks at that first, whatever else the block's status is
         (expand.c:725-726), then at the status, and at the CRC only of a block that is OK (:730-733) */
      if (kernel_ok && b.nblock > level * 100000u) { b.err = 8; behind_the_block = true; }
      if (b.err) {
        char buf[128];
        snprintf(buf, sizeof buf, "lbzamd_decompress: block %u: %s (code %u)", nblocks, b.err == 11 ? "CRC mismatch" : "malformed block", b.err);
        /* What the reference's PROGRAM says (lbzamd_last_error_code).  Its parser walks the chain ahead of everything else and
           fails at once on what IT finds (expand.c:395-491: a missing magic, a stream CRC that is not the fold of the stored
           block CRCs, the end of the file), while a block's own status is reported only when the muxer reaches the block
           (:726-735).  So (a) a block that breaks off inside its tables or codes leaves the parser at the bit where
           retrieve() stopped, and what it finds there is not a block magic: ERR_HEADER; (b) an error noticed behind the
           block's last code (CRC, size against the stream's level, origin pointer, empty block) is remembered -- the first such block
           of the chain --, the walk goes on, and it is reported only if the parser finds nothing of its own further down (the
           reference's decompressor suite: crc2). */
        if (behind_the_block) {                                  /* (origin pointer behind the block, empty block: retrieve() has taken the whole block by then) */
          if (!pend_code) { pend_code = dec_error((int)b.err, b.nblock); pend_msg = buf; pend_total = total; }
        } else {
          g_err = buf;
          /* (c) a block whose codes ran past the last byte of the file (what it read there were zeros): retrieve() asked
             for more input and there was none -- ERR_EOF (decode.c:393-399) */
          g_err_code = b.bit_used > padbits ? RE_EOF : RE_HEADER;
          stop = true;
          break;
        }
      }
      if (pend_code) { b.err = 99; b.out_len = 0; }            /* the walk goes on (the parser may still find something of its own); no bytes from here on */
      else taken[ci] = 1;
      b.out_off = total;
      total += b.out_len;
      cc = ((cc << 1) | (cc >> 31)) ^ b.stored_crc;            /* encode.h:38 written for the inverted values */
      expect = b.bit_used;
      nblocks++; stream_blocks++;
    }
    if (!stop && !carry && last_batch && in_stream && near_end(expect)) carry = true;      /* the next magic is not (all) here yet */
    if (!stop && !carry && last_batch && in_stream) {
      g_err_code = no_magic_at(expect);
      g_err = g_err_code == RE_EOF ? "lbzamd_decompress: stream without end-of-stream marker (truncated?)"
                                   : "lbzamd_decompress: no block or end-of-stream magic where the previous block ends (damaged or overrun block)";
      stop = true;
    }
    for (u32 i = 0; i < nb; i++)                               /* what the walk did not take: off the chain, behind an error, the refused block itself */
      if (!taken[b0 + i]) { hb[b0 + i].err = hb[b0 + i].err ? hb[b0 + i].err : 99u; hb[b0 + i].out_len = 0; }
    const uint64_t emit_total = pend_code ? pend_total : total;   /* bytes of the blocks taken so far */
    if (c->grow_out && nb && emit_total > out_cap) {
      /* the size is known only now (the blocks of this pass are decoded, their bytes not yet in place): a larger buffer,
         sized for the passes still to come as the blocks so far suggest, keeps what the earlier passes have written */
      const uint64_t prev = emitted;                           /* (not hb[b0].out_off: the batch's first candidate may be off the chain -- a
                                                                  magic inside a payload -- and never get an offset) */
      uint64_t want = emit_total + emit_total / 16u + 4096u;
      if (b0 + nb < hb.size()) want = (uint64_t)((double)emit_total * (double)hb.size() / (double)(b0 + nb) * 1.0625) + 4096u;
      u8 *bigger = nullptr;
      HIPCHK(hipMalloc((void **)&bigger, want + 256u));
      if (prev) HIPCHK(hipMemcpyAsync(bigger, c->d_out, prev, hipMemcpyDeviceToDevice, q));
      HIPCHK(hipStreamSynchronize(q));
      (void)hipFree(c->d_out);
      c->d_out = bigger; c->d_out_cap = want;
      d_out = bigger; out_cap = want;
    }
    if (nb && emit_total <= out_cap && d_out && emit_total > emitted) {
      HIPCHK(hipMemcpyAsync(c->blocks, hb.data() + b0, nb * sizeof(lbz_dblock), hipMemcpyHostToDevice, q));
      HIPCHK(hipEventRecord(c->ev[5], q));
      hipLaunchKernelGGL(k_demit, dim3(nb * 8u), dim3(256), 0, q, (const lbz_dblock *)c->blocks, nb, (const u8 *)c->W, (const u32 *)c->pinfo, d_out, (u64)out_cap, c->cap);
      HIPCHK(hipEventRecord(c->ev[6], q));
      HIPCHK(hipStreamSynchronize(q));
      HIPCHK(hipGetLastError());
      float t = 0;
      HIPCHK(hipEventElapsedTime(&t, c->ev[5], c->ev[6])); ms[4] += t;
      emitted = emit_total;
    }
    if (stop) { *out_len = (size_t)emitted; return -3; }
    if (hb.empty() || carry) break;
  }
  c->stats.nblocks = nblocks;
  c->stats.nstreams = (uint32_t)trailers.size();
  /* 5. stream CRCs: the fold of the accepted blocks' stored CRCs against each trailer */
  {
    std::vector<uint8_t> tail(8);
    for (const trailer &tr : trailers) {
      const uint64_t by = (tr.bit + 48) >> 3;
      const size_t nbytes = std::min<size_t>(8, len - by);
      HIPCHK(hipMemcpy(tail.data(), d_in + by, nbytes, hipMemcpyDeviceToHost));
      std::vector<uint8_t> h(tail.begin(), tail.begin() + nbytes);
      if (tr.bit + 80u > padbits) { g_err = "lbzamd_decompress: the end-of-stream marker's CRC is cut off"; g_err_code = RE_EOF; *out_len = (size_t)emitted; return -3; }   /* parse.c:276 */
      const uint32_t want = rd_be32_bits(h, (tr.bit + 48) & 7u);
      if (want != tr.cc) { g_err = "lbzamd_decompress: stream CRC mismatch"; g_err_code = RE_STRMCRC; *out_len = (size_t)emitted; return -3; }
    }
  }
  if (pend_code) { g_err = pend_msg; g_err_code = pend_code; *out_len = (size_t)emitted; return -3; }
  if (rs) {
    /* where the next window begins: in a stream at the magic the chain expects next; between streams (a marker whose sequel
       was cut off cannot be here: it is carried as a whole) at the byte behind the last trailer */
    rs->consumed_bit = finished ? lenbits : expect;
    rs->in_stream = in_stream ? 1u : 0u; rs->level = level; rs->cc = cc; rs->stream_blocks = stream_blocks; rs->finished = finished ? 1u : 0u;
    rs->started = 1u;
    rs->base_bytes = base_bytes + rs->consumed_bit / 8u;
    rs->nblocks_total += nblocks; rs->nstreams_total += (uint32_t)trailers.size();
  }
  c->stats.n_out = total;
  c->stats.ms_scan = ms[0]; c->stats.ms_huff = ms[1]; c->stats.ms_sort = ms[2]; c->stats.ms_walk = ms[3]; c->stats.ms_emit = ms[4];
  c->stats.ms_blocks = ms[5];
  c->stats.ms_total = ms[0] + ms[5] + ms[4];
  *out_len = (size_t)total;
  if (total > out_cap || (total && !d_out)) { g_err = "lbzamd_decompress: output buffer too small"; return -2; }
  return 0;
}

extern "C" int lbzamd_decompress_host(lbzamd_dctx *c, const uint8_t *in, size_t len, uint8_t *out, size_t out_cap, size_t *out_len)
{
  if (!c || !out_len || (len && !in)) { g_err = "lbzamd_decompress_host: bad argument"; return -1; }
  HIPCHK(hipSetDevice(c->device));
  if (len + 16 > c->d_in_cap) {
    (void)hipFree(c->d_in); c->d_in = nullptr; c->d_in_cap = 0;
    HIPCHK(hipMalloc((void **)&c->d_in, len + 256));
    c->d_in_cap = len + 16;
  }
  if (out_cap > c->d_out_cap) {
    (void)hipFree(c->d_out); c->d_out = nullptr; c->d_out_cap = 0;
    HIPCHK(hipMalloc((void **)&c->d_out, out_cap + 256));
    c->d_out_cap = out_cap;
  }
  HIPCHK(hipMemcpy(c->d_in, in, len, hipMemcpyHostToDevice));
  const int rc = lbzamd_decompress_device(c, c->d_in, len, out_cap ? c->d_out : nullptr, out_cap, out_len);
  if (rc && rc != -3) return rc;
  if (*out_len && *out_len <= out_cap) HIPCHK(hipMemcpy(out, c->d_out, *out_len, hipMemcpyDeviceToHost));   /* (-3: the bytes in front of the error) */
  else if (rc) *out_len = 0;
  return rc;
}

/* One pass for callers that do not know the decoded size: the device output buffer grows between the block passes, the
 * result comes back in a malloc'ed buffer (lbzamd_free).  lbzamd_decompress_host with a buffer that turns out too small
 * has decoded every block by the time it knows, and the second call decodes them again.                              */
extern "C" int lbzamd_decompress_alloc(lbzamd_dctx *c, const uint8_t *in, size_t len, uint8_t **out, size_t *out_len)
{
  if (!c || !out || !out_len || (len && !in)) { g_err = "lbzamd_decompress_alloc: bad argument"; return -1; }
  HIPCHK(hipSetDevice(c->device));
  *out = nullptr; *out_len = 0;
  if (len + 16 > c->d_in_cap) {
    (void)hipFree(c->d_in); c->d_in = nullptr; c->d_in_cap = 0;
    HIPCHK(hipMalloc((void **)&c->d_in, len + 256));
    c->d_in_cap = len + 16;
  }
  if (!c->d_out_cap) {
    const size_t guess = 4u * len + 65536u;
    HIPCHK(hipMalloc((void **)&c->d_out, guess + 256));
    c->d_out_cap = guess;
  }
  HIPCHK(hipMemcpy(c->d_in, in, len, hipMemcpyHostToDevice));
  c->grow_out = true;
  size_t n = 0;
  const int rc = lbzamd_decompress_device(c, c->d_in, len, c->d_out, c->d_out_cap, &n);
  c->grow_out = false;
  if (rc && (rc != -3 || n == 0)) return rc;
  const std::string why = rc ? g_err : std::string();            /* (-3 with n > 0: the bytes in front of the error come back too) */
  const int why_code = g_err_code;
  uint8_t *h = (uint8_t *)malloc(n ? n : 1);
  if (!h) { if (rc) return rc; g_err = "lbzamd_decompress_alloc: out of host memory"; return -1; }
  if (n) {
    const hipError_t e = hipMemcpy(h, c->d_out, n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(h); if (rc) { g_err = why; g_err_code = why_code; return rc; } return fail_msg("hipMemcpy", e); }
  }
  *out = h; *out_len = n;
  return rc;
}
extern "C" void lbzamd_free(void *p) { free(p); }

/* one window of a longer input: include/lbzip2_amd.h */
extern "C" int lbzamd_decompress_window(lbzamd_dctx *c, const uint8_t *in, size_t len, int final, lbzamd_dresume *rs, uint8_t **out, size_t *out_len)
{
  if (!c || !rs || !out || !out_len) { g_err = "lbzamd_decompress_window: bad argument"; return -1; }
  c->rs = rs; c->rs_final = final != 0;
  const int rc = lbzamd_decompress_alloc(c, in, len, out, out_len);
  c->rs = nullptr; c->rs_final = true;
  return rc;
}

extern "C" int lbzamd_dget_stats(lbzamd_dctx *c, lbzamd_dstats *st)
{
  if (!c || !st) return -1;
  *st = c->stats;
  return 0;
}

/* ===================================================================== drop-in (A) */
/* The reference calls collect / encode / transmit once per block, from many worker threads at
 * once (compress.c:81-115 runs them outside the scheduler lock).  One block per launch would
 * leave the device empty -- a kernel of one workgroup, and a handful of hardware queues for any
 * number of streams -- so the calls of all threads are COMBINED: every state leases one slab of a
 * shared pool context; a call posts a request and the first thread to find no leader becomes the
 * leader, takes every request posted so far (collects and encodes alike), runs them as one round
 * of launches over the listed slabs, publishes the block records and wakes the others.  With N
 * worker threads a round has up to N blocks, as in the batch interface.
 *
 * encoder_state as seen by the caller: an opaque blob it malloc'ed.  We keep a small header in
 * it; the compressed block is staged after the header for transmit(NULL).                   */
struct wu_pool;
struct encoder_state {
  uint32_t magic;
  uint32_t mbs;
  uint32_t cf;
  uint32_t out_len;
  uint32_t crc;
  uint32_t collected;
  wu_pool *pool;              /* set while the state holds a slab */
  uint32_t slab;
  uint32_t seq_cap;           /* re-entered collect() (the reference's -u mode): the block's raw bytes so far live in */
  u8 *seq_dev;                /* a device buffer of their own, seq_cap bytes                                         */
  uint32_t seq_full;          /* the block is full: further collect() calls take nothing */
  uint32_t pad_[3];
};
#define ENC_MAGIC 0x6c627a41u

[[noreturn]] static void die(const char *what)
{
  fprintf(stderr, "lbzip2_amd: fatal: %s: %s\n", what, g_err.c_str());
  abort();
}
#define HIPDIE(x, what) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); die(what); } } while (0)

struct wu_req {
  uint32_t slab, len;
  const uint8_t *buf;         /* collect: the caller's bytes (it is blocked until the round is done) */
  int stage;                  /* 0 = collect, 1 = encode */
  bool done;
  uint32_t consumed, out_len, crc, err;       /* the block record as of the round that served the request */
  std::condition_variable cv; /* the request's own: its thread is woken when the request is done, or to lead a round -- and no
                                 other thread with it.  (One condition variable for the pool, broadcast at the end of every
                                 round, kept 47 of 64 caller threads runnable at any moment, in the kernel, on the futex: 14
                                 CPUs of system time, and the box's CPU quota throttled the process for 68 ms of every 100:
                                 profiles/r06_i_wu_*.txt) */
};

struct wu_lane {                      /* a stream that runs one round at a time, with the round's lists and (encode) its own set of BWT workspaces */
  hipStream_t q = nullptr;            /* non-blocking: no ties to the null stream */
  u32 *d_list = nullptr, *d_len = nullptr;
  u32 *h_pick = nullptr;              /* pinned: {consumed, out_len, crc, err} per request of the round; the device writes it */
  u32 *h_list = nullptr;              /* pinned: the round's slabs, then their lengths (the source of the list copies) */
  u8 *ws = nullptr;                   /* encode lanes: nslots slots of the pool context's workspace */
  bool busy = false;
};
#define WU_COLLECT_LANES 2u
#define WU_ENCODE_LANES 3u

struct wu_pool {
  lbzamd_ctx *c = nullptr;    /* P resident slabs, staging for P slabs of input */
  uint32_t P = 0;
  /* Requests of one kind (0 = collect, 1 = encode) wait in one queue; whoever finds a free lane of that kind leads a round
     on it.  Several rounds of a kind run side by side (round 6): with one encode round at a time the callers moved in
     convoy -- the first block back from a round went through collect alone and then held the only encode lane for a
     round of ONE block (7 ms of launch chain) while the other 63 waited: 2.85 GB/s at 64 threads (profiles/r05_workunits.txt). */
  std::vector<wu_req *> pending[2];
  std::vector<wu_lane> lanes[2];
  uint32_t inflight[2] = { 0, 0 };        /* rounds running, per kind */
  bool gathering[2] = { false, false };   /* a leader is waiting for stragglers: arrivals join its round */
  u8 *h_in = nullptr, *h_out = nullptr;   /* pinned staging, one slab each: callers fill / drain it in parallel, the leader's copies are pure DMA */
  std::mutex mu;
  std::condition_variable cv_slab, cv_arrive; /* a slab was freed (states beyond the pool's slabs wait here); a request arrived (a gathering leader listens) */
  std::vector<uint32_t> free_slabs;
  /* re-entered collect(): one at a time (the reference holds a token, compress.c:62,143), own stream and scratch */
  std::mutex seq_mu;
  hipStream_t seq_q = nullptr;
  unsigned long long *seq_starts = nullptr;
  u32 *seq_ticket = nullptr;
  lbz_seq_out *seq_out = nullptr;
  std::vector<std::pair<u8 *, uint32_t>> seq_free;     /* buffers of finished blocks, kept for the next ones */
};

/* One pool per (device, block size).  LBZAMD_DEVICES = N ("all": every device; default 1) spreads the states over
 * N devices: a state leases its slab from the pool of device (lease counter mod N), so the worker threads of the
 * reference's unmodified pipeline (process.c:515-548) keep every GPU's rounds full.                            */
#define WU_MAX_DEV 16
static std::mutex g_pools_mu;
static wu_pool *g_pools[WU_MAX_DEV][10];
static int g_pool_devs = 0;                  /* 0 = not read yet */
static unsigned g_pool_next = 0;

static wu_pool *pool_on(int device, unsigned bs100k);
static wu_pool *pool_for(unsigned bs100k)
{
  int dev = -1;
  {
    std::lock_guard<std::mutex> lk(g_pools_mu);
    if (!g_pool_devs) {
      const char *env = getenv("LBZAMD_DEVICES");
      const int have = logical_devices();
      if (have < 1) { g_err = "no HIP device (this library has no CPU path)"; die("work-unit pool"); }
      int want = env ? (!strcmp(env, "all") ? have : atoi(env)) : 1;
      if (want < 1) want = 1;
      if (want > have) want = have;
      if (want > WU_MAX_DEV) want = WU_MAX_DEV;
      g_pool_devs = want;
    }
    if (g_pool_devs > 1) dev = (int)(g_pool_next++ % (unsigned)g_pool_devs);
  }
  return pool_on(dev, bs100k);
}

static wu_pool *pool_on(int device, unsigned bs100k)
{
  std::lock_guard<std::mutex> lk(g_pools_mu);
  if (device < 0) HIPDIE(hipGetDevice(&device), "work-unit pool");
  if (device >= WU_MAX_DEV) device = WU_MAX_DEV - 1;
  if (g_pools[device][bs100k]) return g_pools[device][bs100k];
  wu_pool *p = new wu_pool;
  const char *env = getenv("LBZAMD_POOL_SLABS");
  p->P = env ? (uint32_t)atoi(env) : 256u;      /* states in flight per pool: 256 slabs are 16 GB of HBM and 0.5 GB of page-locked memory at -9
                                                   (1024, the default until round 4: 35 GB and 2 GB on the first collect() whatever the number of
                                                   threads); callers beyond that wait for a slab */
  if (p->P < 1u) p->P = 1u;
  /* a slot set per encode lane, each for half the pool's slabs (a bigger round goes through its lane in pieces) */
  const uint32_t lane_slots = p->P <= 2u ? p->P : (p->P + 1u) / 2u;
  if (ctx_create(&p->c, device, bs100k, p->P + 1u, lane_slots, p->P <= 2u ? 1u : WU_ENCODE_LANES)) die("cannot create the work-unit pool");      /* (+ 1: the spare slab, always empty: k_pool_split) */
  lbzamd_ctx *c = p->c;
  if (ensure_staging(c, (size_t)p->P * c->L.M, 0)) die("work-unit pool staging");
  HIPDIE(hipMemset(c->meta + 2u * (size_t)p->P, 0, 2u * sizeof(lbz_block_meta)), "pool");
  p->lanes[0].resize(WU_COLLECT_LANES);
  p->lanes[1].resize(c->nstreams);                  /* a lane per slot set */
  if (getenv("LBZAMD_POOL_LANES") && atoi(getenv("LBZAMD_POOL_LANES")) >= 1 && (size_t)atoi(getenv("LBZAMD_POOL_LANES")) < p->lanes[1].size()) {
    p->lanes[1].resize((size_t)atoi(getenv("LBZAMD_POOL_LANES")));      /* (tuning) */
    p->lanes[0].resize(1);
  }
  for (int k = 0; k < 2; k++)
    for (size_t i = 0; i < p->lanes[k].size(); i++) {
      wu_lane &l = p->lanes[k][i];
      HIPDIE(hipStreamCreateWithFlags(&l.q, hipStreamNonBlocking), "pool");
      HIPDIE(hipMalloc((void **)&l.d_list, p->P * sizeof(u32)), "pool");
      HIPDIE(hipMalloc((void **)&l.d_len, 2u * p->P * sizeof(u32)), "pool");
      HIPDIE(hipHostMalloc((void **)&l.h_pick, p->P * 4u * sizeof(u32), hipHostMallocPortable), "pool");
      HIPDIE(hipHostMalloc((void **)&l.h_list, p->P * 2u * sizeof(u32), hipHostMallocPortable), "pool");
      if (k == 1) l.ws = c->ws + (size_t)(i % c->nstreams) * c->nslots * (c->slot_bytes + c->spill_bytes);
    }
  HIPDIE(hipHostMalloc((void **)&p->h_in, (size_t)p->P * c->L.M, hipHostMallocPortable), "pool");
  HIPDIE(hipHostMalloc((void **)&p->h_out, (size_t)p->P * c->L.out_a, hipHostMallocPortable), "pool");
  for (uint32_t i = p->P; i-- > 0;) p->free_slabs.push_back(i);
  HIPDIE(hipStreamCreateWithFlags(&p->seq_q, hipStreamNonBlocking), "pool");
  HIPDIE(hipMalloc((void **)&p->seq_starts, 4u * sizeof(unsigned long long)), "pool");
  HIPDIE(hipMalloc((void **)&p->seq_ticket, sizeof(u32)), "pool");
  HIPDIE(hipMalloc((void **)&p->seq_out, sizeof(lbz_seq_out)), "pool");
  g_pools[device][bs100k] = p;
  return p;
}

/* One round of one lane.  Collect rounds (H2D of the callers' slabs, k_collect) and encode rounds (the BWT,
 * MTF and coding kernels over the listed blocks, then the packed blocks D2H) run on their own streams with
 * their own leaders: a collect round touches only slabs whose states are in collect(), an encode round only
 * slabs whose states are in encode(), so the two overlap -- while the device works through an encode round, the
 * callers whose transmit() returned are already collecting, and the next encode round is full when this one ends. */
static void pool_round(wu_pool *p, int stage, wu_lane &ln, std::vector<wu_req *> &batch)
{
  lbzamd_ctx *c = p->c;
  static const bool trace = getenv("LBZAMD_POOL_TRACE") != nullptr;
  const double t0 = trace ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
  HIPDIE(hipSetDevice(c->device), "work-unit round");
  const u32 cnt = (u32)batch.size();
  for (u32 i = 0; i < cnt; i++) { ln.h_list[i] = batch[i]->slab; ln.h_list[cnt + i] = batch[i]->len; }
  HIPDIE(hipMemcpyAsync(ln.d_list, ln.h_list, cnt * sizeof(u32), hipMemcpyHostToDevice, ln.q), "round");
  if (stage == 0) {
    HIPDIE(hipMemcpyAsync(ln.d_len, ln.h_list + cnt, cnt * sizeof(u32), hipMemcpyHostToDevice, ln.q), "collect");
    /* the callers' slabs: read from the page-locked staging area by the device itself (k_pool_in) -- one launch, not a copy
       call per slab */
    const u32 parts = cnt >= 128u ? 4u : (cnt >= 32u ? 8u : 16u);
    hipLaunchKernelGGL(k_pool_in, dim3(parts * cnt), dim3(256), 0, ln.q, (const u8 *)p->h_in, c->d_in, c->L.M, (const u32 *)ln.d_list, (const u32 *)ln.d_len, parts);
    hipLaunchKernelGGL(k_collect, dim3(cnt), dim3(LBZ_COLLECT_WG), 0, ln.q, (const u8 *)c->d_in,
                       (u64)p->P * c->L.M, c->L, c->T, c->meta, 0u, (const u32 *)ln.d_list, (const u32 *)ln.d_len);
    hipLaunchKernelGGL(k_meta_pick, dim3((cnt + 255u) / 256u), dim3(256), 0, ln.q, (const lbz_block_meta *)c->meta, (const u32 *)ln.d_list, cnt, ln.h_pick);
  } else {
    u8 *ws = ln.ws, *wsp = ln.ws + (size_t)c->nslots * c->slot_bytes;
    const bool beside = p->lanes[1].size() > 1u;              /* other rounds may run beside this one */
    /* A block the text rounds do not finish (a repeat-heavy one: 17 rank rounds, 60 ms for a lone block) must not hold up the
       round's other blocks, whose callers wait: behind the text rounds the round's list is split in two of the same length
       (k_pool_split: the absent entries name the pool's spare slab, whose blocks are empty) -- the blocks that are sorted go
       through MTF, coding and k_pool_out at once and their callers are released; if any block is left, the rank rounds and
       the same three stages follow for those. */
    u32 *fast = ln.d_len, *slow = ln.d_len + cnt;             /* (d_len: free in an encode round; 2 P entries) */
    for (u32 i = 0; i < cnt; i++) ln.h_pick[4u * i + 3u] = 0xFFFFFFFFu;          /* "not back yet" */
    for (u32 o = 0; o < cnt; o += c->nslots) {
      /* primaries only (grid = count): what collect() left over went back to the caller */
      const u32 count = cnt - o < c->nslots ? cnt - o : c->nslots;
      const u32 *lst = ln.d_list + o;
      for (int ph = 0; ph < 2; ph++) launch_sort(c, ln.q, 0u, count, count, ws, wsp, lst, ph, beside);
      launch_sort(c, ln.q, 0u, count, count, ws, wsp, lst, 3, beside);
      hipLaunchKernelGGL(k_pool_split, dim3((count + 255u) / 256u), dim3(256), 0, ln.q, (const lbz_block_meta *)c->meta, lst, count, fast + o, slow + o, p->P);
      launch_mtf(c, ln.q, 0u, count, count, count, fast + o);
      hipLaunchKernelGGL(k_encode, dim3(count), dim3(LBZ_ENCODE_WG), 0, ln.q, (const u16 *)c->V, (const u32 *)c->freq, c->O, c->meta, c->L, 0u, count, (const u32 *)(fast + o));
      if (cnt > c->nslots) {                                   /* (a round in pieces: the next piece needs the workspaces) */
        launch_sort(c, ln.q, 0u, count, count, ws, wsp, slow + o, 4, beside);
        launch_mtf(c, ln.q, 0u, count, count, count, slow + o);
        hipLaunchKernelGGL(k_encode, dim3(count), dim3(LBZ_ENCODE_WG), 0, ln.q, (const u16 *)c->V, (const u32 *)c->freq, c->O, c->meta, c->L, 0u, count, (const u32 *)(slow + o));
      }
    }
    /* the packed blocks and the four words each caller waits for: written into the page-locked staging area by the device */
    const u32 oparts = cnt >= 64u ? 2u : 4u;
    if (cnt > c->nslots) {
      hipLaunchKernelGGL(k_pool_out, dim3(oparts * cnt), dim3(256), 0, ln.q, (const u8 *)c->O, (const lbz_block_meta *)c->meta, c->L,
                         (const u32 *)ln.d_list, p->h_out, ln.h_pick, oparts, p->P);
    } else {
      hipLaunchKernelGGL(k_pool_out, dim3(oparts * cnt), dim3(256), 0, ln.q, (const u8 *)c->O, (const lbz_block_meta *)c->meta, c->L,
                         (const u32 *)fast, p->h_out, ln.h_pick, oparts, p->P);
      /* the callers whose blocks are done; the rank rounds and the rest only if a block needs them */
      HIPDIE(hipStreamSynchronize(ln.q), "round");
      u32 back = 0;
      for (u32 i = 0; i < cnt; i++)
        if (ln.h_pick[4u * i + 3u] != 0xFFFFFFFFu) back++;
      if (back < cnt) {
        launch_sort(c, ln.q, 0u, cnt, cnt, ws, wsp, slow, 4, beside);          /* (the split list: the released callers' slabs may be in other rounds by now) */
        launch_mtf(c, ln.q, 0u, cnt, cnt, cnt, slow);
        hipLaunchKernelGGL(k_encode, dim3(cnt), dim3(LBZ_ENCODE_WG), 0, ln.q, (const u16 *)c->V, (const u32 *)c->freq, c->O, c->meta, c->L, 0u, cnt, (const u32 *)slow);
        hipLaunchKernelGGL(k_pool_out, dim3(oparts * cnt), dim3(256), 0, ln.q, (const u8 *)c->O, (const lbz_block_meta *)c->meta, c->L,
                           (const u32 *)slow, p->h_out, ln.h_pick, oparts, p->P);
        if (back) {
          std::lock_guard<std::mutex> lk(p->mu);
          for (u32 i = 0; i < cnt; i++)
            if (ln.h_pick[4u * i + 3u] != 0xFFFFFFFFu) {
              wu_req *b = batch[i];
              b->consumed = ln.h_pick[4u * i]; b->out_len = ln.h_pick[4u * i + 1u]; b->crc = ln.h_pick[4u * i + 2u]; b->err = ln.h_pick[4u * i + 3u];
              b->done = true;
              b->cv.notify_one();
              batch[i] = nullptr;                             /* its thread goes on: the request is gone */
            }
        }
      }
    }
  }
  HIPDIE(hipStreamSynchronize(ln.q), "round");
  HIPDIE(hipGetLastError(), "round");
  for (u32 i = 0; i < cnt; i++) {
    wu_req *b = batch[i];
    if (!b) continue;                                          /* released early: its thread has gone on */
    b->consumed = ln.h_pick[4u * i]; b->out_len = ln.h_pick[4u * i + 1u]; b->crc = ln.h_pick[4u * i + 2u]; b->err = ln.h_pick[4u * i + 3u];
  }
  if (trace) {
    const double t1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(stderr, "round kind %d lane %d blocks %u at %.2f ms took %.2f ms\n", stage, (int)(&ln - &p->lanes[stage][0]), cnt, t0 - 1e3 * (double)(long)(t0 / 1e3), t1 - t0);
  }
}

/* Post a request and return when it is done.  A waiting request's thread leads a round when a lane of its kind is free and
 * the round is worth starting: a collect round at once (it is short; whoever arrives meanwhile forms the next one), an encode
 * round when the device has no encode round to work on -- after a pause of a few hundred microseconds for the rest of the
 * burst the last collect round released, for a round of two blocks costs the launch chain of a round of sixty -- or when
 * HALF of the states that hold a slab are waiting for one: the callers then move as two convoys whose rounds overlap on the
 * device, one in the kernels while the other is in transmit(), the caller's own code and collect().  The others sleep until
 * their request is marked done. */
/* may a round of kind k start now, and on which lane (mu held) */
static wu_lane *pool_may_lead(wu_pool *p, int k)
{
  if (p->pending[k].empty() || p->gathering[k]) return nullptr;
  const uint32_t in_use = p->P - (uint32_t)p->free_slabs.size();
  if (!(k == 0 || p->inflight[1] == 0u || p->pending[1].size() * 2u >= in_use)) return nullptr;
  for (wu_lane &l : p->lanes[k]) if (!l.busy) return &l;
  return nullptr;
}
/* something changed that may allow a round (a lane is free again, fewer states hold slabs): wake ONE waiting request's thread
   per kind to lead it (mu held) */
static void pool_kick(wu_pool *p)
{
  for (int k = 0; k < 2; k++)
    if (pool_may_lead(p, k)) p->pending[k].front()->cv.notify_one();
}

static void pool_submit(wu_pool *p, wu_req *r)
{
  static const int gather_us = getenv("LBZAMD_POOL_GATHER") ? atoi(getenv("LBZAMD_POOL_GATHER")) : 100;    /* (tuning) */
  std::unique_lock<std::mutex> lk(p->mu);
  const int k = r->stage;
  r->done = false;
  p->pending[k].push_back(r);
  if (p->gathering[k]) p->cv_arrive.notify_one();
  while (!r->done) {
    wu_lane *free_lane = pool_may_lead(p, k);
    if (!free_lane) { r->cv.wait_for(lk, std::chrono::milliseconds(20)); continue; }      /* (the time limit: a lost wake-up costs a pause, not the job) */
    wu_lane &ln = *free_lane;
    ln.busy = true;
    if (k == 1 && p->inflight[1] == 0u && gather_us > 0) {
      p->gathering[k] = true;
      for (int waits = 0; waits < 6; waits++) {               /* as long as requests keep arriving */
        const size_t had = p->pending[k].size();
        const uint32_t in_use = p->P - (uint32_t)p->free_slabs.size();
        if (had * 2u >= in_use) break;
        p->cv_arrive.wait_for(lk, std::chrono::microseconds(gather_us));
        if (p->pending[k].size() == had) break;
      }
      p->gathering[k] = false;
    }
    std::vector<wu_req *> batch;
    batch.swap(p->pending[k]);
    p->inflight[k]++;
    lk.unlock();
    pool_round(p, k, ln, batch);
    lk.lock();
    for (wu_req *b : batch)
      if (b) { b->done = true; if (b != r) b->cv.notify_one(); }
    p->inflight[k]--;
    ln.busy = false;
    pool_kick(p);
  }
}

extern "C" size_t lbzamd_encoder_alloc_size(unsigned long mbs)
{
  /* header + room for the compressed block (transmit(NULL)); cf. encode.c:108-114 */
  return sizeof(encoder_state) + (size_t)mbs + mbs / 8u + 8192u;
}

extern "C" void lbzamd_encoder_init(encoder_state *e, unsigned long mbs, unsigned cf)
{
  if (!e || mbs == 0 || mbs > LBZ_MAX_BLOCK || mbs % 100000u || cf != LBZ_CLUSTER) {
    g_err = "block size must be k*100000 (1<=k<=9) and cluster factor 8"; die("encoder_init");
  }
  memset(e, 0, sizeof *e);
  e->magic = ENC_MAGIC; e->mbs = (uint32_t)mbs; e->cf = cf;
}

/* collect() called again on a state that holds bytes already (compress.c:160-170, the -u mode: one encoder keeps
 * collecting slab after slab until its block is full).  The block's raw bytes so far move to a device buffer of
 * their own, the new ones are appended, and the block is tokenised again from its start (k_collect_seq with a chain
 * of one): how much of the new input fits is the cut position minus what was there before.  The reference allows
 * one such caller at a time (its collect token); so does this. */
static int collect_again(encoder_state *e, const uint8_t *buf, size_t *buf_sz)
{
  wu_pool *p = e->pool;
  lbzamd_ctx *c = p->c;
  if (e->seq_full || *buf_sz == 0) return e->seq_full ? 1 : 0;
  std::lock_guard<std::mutex> lk(p->seq_mu);
  HIPDIE(hipSetDevice(c->device), "collect");
  const uint32_t have = e->collected;
  const size_t room = (size_t)e->mbs * 52u + 4096u;              /* a block never takes more raw bytes than this */
  const size_t len = *buf_sz < room ? *buf_sz : room;
  const size_t need = (size_t)have + len + 64u;
  if (!e->seq_dev || e->seq_cap < need) {
    u8 *nb = nullptr;
    uint32_t ncap = 0;
    for (size_t i = 0; i < p->seq_free.size(); i++)
      if (p->seq_free[i].second >= need) { nb = p->seq_free[i].first; ncap = p->seq_free[i].second; p->seq_free.erase(p->seq_free.begin() + (long)i); break; }
    if (!nb) {
      size_t want = (size_t)e->mbs * 5u / 2u + 4096u;
      while (want < need) want *= 2u;
      HIPDIE(hipMalloc((void **)&nb, want), "collect");
      ncap = (uint32_t)want;
    }
    const u8 *old = e->seq_dev ? e->seq_dev : c->d_in + (size_t)e->slab * c->L.M;
    if (have) HIPDIE(hipMemcpyAsync(nb, old, have, hipMemcpyDeviceToDevice, p->seq_q), "collect");
    if (e->seq_dev) { HIPDIE(hipStreamSynchronize(p->seq_q), "collect"); p->seq_free.push_back({ e->seq_dev, e->seq_cap }); }
    e->seq_dev = nb; e->seq_cap = ncap;
  }
  HIPDIE(hipMemcpyAsync(e->seq_dev + have, buf, len, hipMemcpyHostToDevice, p->seq_q), "collect");
  const unsigned long long start0 = 1ull;
  HIPDIE(hipMemsetAsync(p->seq_starts, 0, 4u * sizeof(unsigned long long), p->seq_q), "collect");
  HIPDIE(hipMemsetAsync(p->seq_ticket, 0, sizeof(u32), p->seq_q), "collect");
  HIPDIE(hipMemsetAsync(p->seq_out, 0, sizeof(lbz_seq_out), p->seq_q), "collect");
  HIPDIE(hipMemcpyAsync(p->seq_starts, &start0, sizeof start0, hipMemcpyHostToDevice, p->seq_q), "collect");
  hipLaunchKernelGGL(k_collect_seq, dim3(1), dim3(LBZ_COLLECT_WG), 0, p->seq_q, (const u8 *)e->seq_dev, (u64)(have + len), c->L, c->T, c->meta,
                     1u, p->seq_starts, p->seq_ticket, p->seq_out, e->slab,
                     (const unsigned long long *)nullptr, (const unsigned long long *)nullptr, 0u);
  lbz_seq_out so{};
  HIPDIE(hipMemcpyAsync(&so, p->seq_out, sizeof so, hipMemcpyDeviceToHost, p->seq_q), "collect");
  HIPDIE(hipStreamSynchronize(p->seq_q), "collect");
  HIPDIE(hipGetLastError(), "collect");
  if (so.err || so.next < have) { g_err = "device error while re-entering collect()"; die("collect"); }
  const size_t took = (size_t)so.next - have;
  e->collected = (uint32_t)so.next;
  *buf_sz -= took;
  if (took < len) e->seq_full = 1;                              /* input left over: the block is full (encode.c:335) */
  return e->seq_full ? 1 : 0;
}

extern "C" int lbzamd_collect(encoder_state *e, const uint8_t *buf, size_t *buf_sz)
{
  if (!e || e->magic != ENC_MAGIC || !buf || !buf_sz) { g_err = "bad encoder state"; die("collect"); }
  if (e->pool) return collect_again(e, buf, buf_sz);
  const size_t avail = *buf_sz < e->mbs ? *buf_sz : e->mbs;
  if (avail == 0) return 0;
  wu_pool *p = pool_for(e->mbs / 100000u);
  {
    std::unique_lock<std::mutex> lk(p->mu);
    while (p->free_slabs.empty()) p->cv_slab.wait(lk);      /* more states in flight than the pool has slabs */
    e->slab = p->free_slabs.back();
    p->free_slabs.pop_back();
  }
  e->pool = p;
  memcpy(p->h_in + (size_t)e->slab * p->c->L.M, buf, avail);            /* in the caller's thread */
  wu_req r;
  r.slab = e->slab; r.len = (uint32_t)avail; r.buf = buf; r.stage = 0; r.done = false; r.consumed = r.out_len = r.crc = r.err = 0u;
  pool_submit(p, &r);
  e->collected = r.consumed;
  *buf_sz -= r.consumed;
  if (r.consumed < avail) e->seq_full = 1;                      /* input left over: the block is full */
  return r.consumed < avail;
}

extern "C" size_t lbzamd_encode(encoder_state *e, uint32_t *crc)
{
  if (!e || e->magic != ENC_MAGIC || !e->pool || !crc) { g_err = "encode() before collect()"; die("encode"); }
  wu_pool *p = e->pool;
  wu_req r;
  r.slab = e->slab; r.len = 0u; r.buf = nullptr; r.stage = 1; r.done = false; r.consumed = r.out_len = r.crc = r.err = 0u;
  pool_submit(p, &r);
  if (r.err) { g_err = "device pipeline error"; die("encode"); }
  e->out_len = r.out_len;
  e->crc = r.crc;
  *crc = r.crc;
  return r.out_len;
}

static void pool_release(encoder_state *e)
{
  wu_pool *p = e->pool;
  if (e->seq_dev) {
    std::lock_guard<std::mutex> lk(p->seq_mu);
    if (p->seq_free.size() < 64u) p->seq_free.push_back({ e->seq_dev, e->seq_cap });
    else (void)hipFree(e->seq_dev);
    e->seq_dev = nullptr; e->seq_cap = 0;
  }
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->free_slabs.push_back(e->slab);
    pool_kick(p);                                   /* (one state fewer holds a slab: the waiting ones may be half of them now) */
  }
  p->cv_slab.notify_one();
  e->pool = nullptr;
}

extern "C" void *lbzamd_transmit(encoder_state *e, void *buf)
{
  if (!e || e->magic != ENC_MAGIC || !e->pool) { g_err = "transmit() before encode()"; die("transmit"); }
  wu_pool *p = e->pool;
  lbzamd_ctx *c = p->c;
  if (!buf) buf = (void *)(e + 1);
  const size_t bytes = ((size_t)e->out_len + 3u) / 4u * 4u;          /* whole words, compress.c:220 */
  memcpy(buf, p->h_out + (size_t)e->slab * c->L.out_a, bytes);          /* staged by the round that encoded it */
  pool_release(e);
  return buf;
}

extern "C" void lbzamd_encoder_abandon(encoder_state *e)
{
  if (e && e->magic == ENC_MAGIC && e->pool) pool_release(e);
}

/* the reference's own symbol names (encode.h:29-33) */
extern "C" size_t encoder_alloc_size(unsigned long mbs) { return lbzamd_encoder_alloc_size(mbs); }
extern "C" void encoder_init(encoder_state *e, unsigned long mbs, unsigned cf) { lbzamd_encoder_init(e, mbs, cf); }
extern "C" int collect(encoder_state *e, const uint8_t *buf, size_t *buf_sz) { return lbzamd_collect(e, buf, buf_sz); }
extern "C" size_t encode(encoder_state *e, uint32_t *crc) { return lbzamd_encode(e, crc); }
extern "C" void *transmit(encoder_state *e, void *buf) { return lbzamd_transmit(e, buf); }

/* ===================================================================== (D) decode.h's work-unit interface
 * retrieve / decode / emit on ONE block of one worker thread, as src/expand.c:547-690 drives them.  One block per launch --
 * what the call sequence suggests -- measured 70 MB/s with any number of threads (launches of different host threads'
 * streams do not overlap the way one launch over many blocks does); so the threads' requests are COMBINED, as the encoder's
 * are: the first thread to find no leader decodes every posted block in one k_dblock / k_demit pass and wakes the rest. */
#include <arpa/inet.h>
namespace {
enum {                                    /* src/common.h:54-76 */
  WD_OK = 0, WD_MORE, WD_FINISH, WD_ERR_MAGIC, WD_ERR_HEADER, WD_ERR_BITMAP, WD_ERR_TREES, WD_ERR_GROUPS, WD_ERR_SELECTOR,
  WD_ERR_DELTA, WD_ERR_PREFIX, WD_ERR_INCOMPLT, WD_ERR_EMPTY, WD_ERR_UNTERM, WD_ERR_RUNLEN, WD_ERR_BLKCRC, WD_ERR_STRMCRC,
  WD_ERR_OVERFLOW, WD_ERR_BWTIDX, WD_ERR_EOF
};
struct wd_state {
  std::vector<uint8_t> bits;              /* 4 zero bytes (where k_dblock reads a stored CRC) + the block's bits, first bit in the top of byte 4 */
  uint64_t nbits = 0;                     /* the block's bits gathered so far */
  uint64_t acc = 0;
  unsigned nacc = 0;                      /* bits of acc not yet in `bits` (< 8) */
  std::vector<uint8_t> out;               /* the decoded block */
  size_t pos = 0;                         /* bytes emit() has handed out */
  uint32_t crc = 0;
  bool runlen = false;                    /* the block ends where a run's count should stand: emit() says so (decode.c:1009, 1104) */
};
struct wd_req {
  wd_state *st;
  size_t nbytes;                          /* bytes of st->bits (+ the partial byte) the decoder may read */
  uint8_t tail;                           /* the partial byte, if st->nacc */
  uint64_t avail;                         /* bits there are (the 32 stand-in bits included) */
  lbz_dblock rec;
  bool past = false, done = false;
  std::condition_variable cv;             /* the request's own (cf. wu_req): its thread alone is woken, when the request is done or to lead the next round */
};
struct wd_pool {
  std::mutex mu;
  std::vector<wd_req *> queue;
  bool leader = false;
  lbzamd_dctx *c = nullptr;
};
wd_pool g_wd;
const unsigned WD_BATCH = 256u;
const size_t WD_OUT_BUDGET = (size_t)1 << 30;   /* decoded bytes handed from the device to the callers per pass of a round */

void wd_put(wd_state *st, uint64_t v, unsigned n)       /* n <= 32 bits, right-aligned in v */
{
  st->acc = (st->acc << n) | (v & ((1ull << n) - 1ull));
  st->nacc += n;
  st->nbits += n;
  while (st->nacc >= 8u) { st->bits.push_back((uint8_t)(st->acc >> (st->nacc - 8u))); st->nacc -= 8u; }
}
/* whole words of the caller's buffer (big-endian in memory, decode.c:404) behind the st->nacc pending bits: four bytes a word */
void wd_put_words(wd_state *st, const uint32_t *p, const uint32_t *limit)
{
  const size_t nw = (size_t)(limit - p);
  if (!nw) return;
  const size_t o = st->bits.size();
  st->bits.resize(o + nw * 4u);
  uint8_t *d = st->bits.data() + o;
  const unsigned k = st->nacc;                            /* < 8 */
  const uint64_t keep = (1ull << k) - 1ull;
  uint64_t acc = st->acc & keep;
  for (; p != limit; p++, d += 4) {
    acc = (acc << 32) | (uint64_t)ntohl(*p);
    const uint32_t w = (uint32_t)(acc >> k);
    d[0] = (uint8_t)(w >> 24); d[1] = (uint8_t)(w >> 16); d[2] = (uint8_t)(w >> 8); d[3] = (uint8_t)w;
    acc &= keep;
  }
  st->acc = acc;
  st->nbits += 32u * (uint64_t)nw;
}
int wd_error(uint32_t code, uint32_t nblock)
{
  switch (code) {
    case 1: return WD_ERR_BITMAP;
    case 2: return WD_ERR_TREES;
    case 3: return WD_ERR_SELECTOR;
    case 4: return WD_ERR_DELTA;
    case 5: return WD_ERR_UNTERM;
    case 6: return WD_ERR_PREFIX;
    case 7: case 8: return WD_ERR_OVERFLOW;
    case 13: return WD_ERR_INCOMPLT;
    case 14: return WD_ERR_GROUPS;
    case 9: return nblock == 0u ? WD_ERR_EMPTY : WD_ERR_BWTIDX;
    default: return WD_ERR_PREFIX;
  }
}
}  // namespace
static int dec_error(int code, uint32_t nblock) { return code == 11 ? RE_BLKCRC : (code == 12 ? (int)WD_ERR_RUNLEN : wd_error((uint32_t)code, nblock)); }
namespace {
void wd_grow_host(u8 **p, size_t *cap, size_t want)
{
  if (want <= *cap) return;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr; *cap = 0;
  want = (want + want / 2u + 65536u + 255u) & ~(size_t)255u;
  HIPDIE(hipHostMalloc((void **)p, want, hipHostMallocDefault), "decoder staging");
  *cap = want;
}
void wd_grow_dev(u8 **p, size_t *cap, size_t want)
{
  if (want <= *cap) return;
  (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  want = want + want / 2u + 65536u;
  HIPDIE(hipMalloc((void **)p, want + 256u), "decoder buffers");
  *cap = want;
}
/* one pass over the posted blocks: bits in, k_dblock, the records back; the bytes of those that decoded: k_demit, out */
void wd_round(const std::vector<wd_req *> &batch)
{
  if (!g_wd.c && lbzamd_dcreate(&g_wd.c, -1, WD_BATCH)) die("decoder_init");
  lbzamd_dctx *c = g_wd.c;
  HIPDIE(hipSetDevice(c->device), "retrieve");
  const u32 nb = (u32)batch.size();
  std::vector<size_t> off(nb);
  size_t total = 0;
  for (u32 i = 0; i < nb; i++) { off[i] = total; total += (batch[i]->nbytes + 16u + 15u) & ~(size_t)15u; }
  const size_t recs = (size_t)nb * sizeof(lbz_dblock);
  wd_grow_host(&c->h_in, &c->h_in_cap, total + recs + 256u);
  wd_grow_dev(&c->d_in, &c->d_in_cap, total + 16u);
  lbz_dblock *hrec = reinterpret_cast<lbz_dblock *>(c->h_in + ((total + 255u) & ~(size_t)255u));
  for (u32 i = 0; i < nb; i++) {
    wd_req *r = batch[i];
    u8 *dst = c->h_in + off[i];
    const size_t whole = r->st->bits.size();
    memcpy(dst, r->st->bits.data(), whole);
    if (r->nbytes > whole) dst[whole] = r->tail;
    memset(dst + r->nbytes, 0, ((r->nbytes + 16u + 15u) & ~(size_t)15u) - r->nbytes);
    lbz_dblock rec{};
    rec.bit_start = (u64)off[i] * 8u;
    rec.max_block = LBZ_MAX_BLOCK;
    hrec[i] = rec;
  }
  hipStream_t q = c->q;
  HIPDIE(hipMemcpyAsync(c->d_in, c->h_in, total, hipMemcpyHostToDevice, q), "retrieve");
  HIPDIE(hipMemcpyAsync(c->blocks, hrec, recs, hipMemcpyHostToDevice, q), "retrieve");
  if (nb <= c->ncus) hipLaunchKernelGGL(k_dblock_w, dim3(nb), dim3(1024), 0, q, (const u8 *)c->d_in, (u64)total, c->blocks, nb, c->tt8, c->tt, c->W, c->pinfo, c->X, c->cap);
  else hipLaunchKernelGGL(k_dblock_m, dim3(nb), dim3(512), 0, q, (const u8 *)c->d_in, (u64)total, c->blocks, nb, c->tt8, c->tt, c->W, c->pinfo, c->X, c->cap);
  HIPDIE(hipMemcpyAsync(hrec, c->blocks, recs, hipMemcpyDeviceToHost, q), "retrieve");
  HIPDIE(hipStreamSynchronize(q), "retrieve");
  HIPDIE(hipGetLastError(), "retrieve");
  /* Ran past the bits there are (or stopped on an error within a word of their end, where what it read were pad bits)?
     Then the block is not all here: decode.c's NEED(). */
  size_t outb = 0;
  for (u32 i = 0; i < nb; i++) {
    wd_req *r = batch[i];
    lbz_dblock &rec = hrec[i];
    const uint64_t used = rec.bit_used - (u64)off[i] * 8u;
    const bool emit_level = rec.err == 11u || rec.err == 12u;    /* what emit() finds: retrieve() and decode() are done with the block */
    /* (an error the kernel found with `used` bits taken is the block's own if those bits were all there -- a code is decided by
       its own bits, a header field by the bits it takes; only "no symbol has this code" (6) looks at bits ahead of the cursor) */
    r->past = used > r->avail || (rec.err == 6u && used + 64u > r->avail);
    rec.bit_used = used;
    if (r->past || (rec.err && !emit_level)) { rec.err = rec.err ? rec.err : 99u; rec.out_len = 0; r->rec = rec; continue; }
    r->st->runlen = rec.err == 12u;
    rec.err = 0;                                                 /* (11: the stand-in CRC does not match, of course) */
    rec.out_off = outb;
    outb += rec.out_len;
    r->rec = rec;
  }
  /* The bytes, in passes of at most WD_OUT_BUDGET: a block of 900 000 bytes can hold runs worth 46 MB (the reference's own
     zip-bomb case), so a round of 256 crafted blocks would ask for 12 GB of device and page-locked memory at once; the
     reference hands its bytes out in out_granul pieces (expand.c:712-735).  A pass takes whole blocks, one at least. */
  (void)outb;
  std::vector<lbz_dblock> pass(nb);
  for (u32 b0 = 0; b0 < nb;) {
    size_t sum = 0;
    u32 b1 = b0;
    for (u32 i = 0; i < nb; i++) { pass[i] = hrec[i]; pass[i].out_len = 0; pass[i].err = 99u; }   /* (k_demit leaves a block with err alone) */
    while (b1 < nb) {
      const wd_req *r = batch[b1];
      const size_t len = (r->rec.err == 0u && !r->past) ? r->rec.out_len : 0u;
      if (b1 > b0 && sum + len > WD_OUT_BUDGET) break;
      if (len) { pass[b1].out_len = (u32)len; pass[b1].out_off = sum; pass[b1].err = 0u; sum += len; }
      b1++;
    }
    if (sum) {
      wd_grow_dev(&c->d_out, &c->d_out_cap, sum);
      wd_grow_host(&c->h_out, &c->h_out_cap, sum + 256u);
      HIPDIE(hipMemcpyAsync(c->blocks, pass.data(), recs, hipMemcpyHostToDevice, q), "retrieve");
      hipLaunchKernelGGL(k_demit, dim3(nb * 8u), dim3(256), 0, q, (const lbz_dblock *)c->blocks, nb, (const u8 *)c->W, (const u32 *)c->pinfo, c->d_out, (u64)c->d_out_cap, c->cap);
      HIPDIE(hipMemcpyAsync(c->h_out, c->d_out, sum, hipMemcpyDeviceToHost, q), "retrieve");
      HIPDIE(hipStreamSynchronize(q), "retrieve");
      HIPDIE(hipGetLastError(), "retrieve");
      for (u32 i = b0; i < b1; i++)
        if (pass[i].out_len) batch[i]->st->out.assign(c->h_out + pass[i].out_off, c->h_out + pass[i].out_off + pass[i].out_len);
    }
    b0 = b1;
  }
}
void wd_submit(wd_req *r)
{
  std::unique_lock<std::mutex> lk(g_wd.mu);
  g_wd.queue.push_back(r);
  for (;;) {
    if (r->done) return;
    if (!g_wd.leader) {
      /* no round under way: this thread runs one over everything posted (its own block among it, as a rule), then lets
         whoever is still waiting lead the next */
      g_wd.leader = true;
      const size_t take = std::min<size_t>(g_wd.queue.size(), WD_BATCH);
      std::vector<wd_req *> batch(g_wd.queue.begin(), g_wd.queue.begin() + take);
      g_wd.queue.erase(g_wd.queue.begin(), g_wd.queue.begin() + take);
      lk.unlock();
      wd_round(batch);
      lk.lock();
      for (wd_req *b : batch) { b->done = true; if (b != r) b->cv.notify_one(); }
      g_wd.leader = false;
      if (!g_wd.queue.empty()) g_wd.queue.front()->cv.notify_one();       /* the next round's leader */
      continue;
    }
    r->cv.wait_for(lk, std::chrono::milliseconds(20));
  }
}
}  // namespace

extern "C" void lbzamd_decoder_init(struct decoder_state *ds)
{
  wd_state *st = new wd_state;
  st->bits.assign(4u, 0u);
  memset(ds, 0, sizeof *ds);
  ds->internal_state = reinterpret_cast<struct retriever_internal_state *>(st);
}

extern "C" void lbzamd_decoder_free(struct decoder_state *ds)
{
  wd_state *st = reinterpret_cast<wd_state *>(ds->internal_state);
  delete st;
  ds->internal_state = nullptr;
}

extern "C" int lbzamd_retrieve(struct decoder_state *ds, struct bitstream *bs)
{
  wd_state *st = reinterpret_cast<wd_state *>(ds->internal_state);
  if (!st || !bs) { g_err = "retrieve(): bad decoder state"; die("retrieve"); }
  /* everything the caller has: the bits left in its buffer word, then whole words (big-endian in memory: decode.c:404) */
  const unsigned live0 = bs->live;
  const uint64_t buff0 = bs->buff;
  const uint32_t *data0 = bs->data;
  const uint64_t before = st->nbits;
  if (live0 > 32u) { wd_put(st, buff0 >> 32, 32u); wd_put(st, (buff0 << 32) >> (64u - (live0 - 32u)), live0 - 32u); }
  else if (live0) wd_put(st, buff0 >> (64u - live0), live0);
  wd_put_words(st, data0, bs->limit);
  /* decode what there is (with whatever the other worker threads have posted) */
  wd_req r{};
  r.st = st;
  r.nbytes = st->bits.size() + (st->nacc ? 1u : 0u);
  r.tail = st->nacc ? (uint8_t)(st->acc << (8u - st->nacc)) : 0u;
  r.avail = 32u + st->nbits;
  wd_submit(&r);
  const lbz_dblock &rec = r.rec;
  if (r.past) {                                                /* everything is taken: MORE, or ERR_EOF at the end of the input */
    bs->live = 0; bs->buff = 0; bs->data = bs->limit;
    if (!bs->eof) return WD_MORE;
    /* no more bits will come: an error the decoder flagged inside the last word is that error, as the reference reports
       it (ERR_PREFIX, ERR_DELTA ...: common.h:54-76), not "unexpected end of file" (99 = it simply ran out of bits) */
    return rec.err && rec.err != 99u ? wd_error(rec.err, rec.nblock) : WD_ERR_EOF;
  }
  /* A malformed block: the error, and the caller's stream taken as consumed -- where it stands means nothing then, and an
     error position in front of THIS call's bits (it lay in the pad bits of the call before and came back as MORE) must
     not be turned into a pointer. */
  /* the caller's stream stands behind the last bit the block took: behind its last code, or -- a malformed block -- where
     retrieve() stopped, as the reference's does (the parser goes on from there and finds what is not a magic: which of the
     two errors the program reports is a race in the reference too) */
  auto stand_behind = [&](uint64_t used) {                        /* `used` bits of THIS call's input */
    if (used <= live0) { bs->buff = used < 64u ? buff0 << used : 0ull; bs->live = live0 - (unsigned)used; bs->data = data0; }
    else {
      const uint64_t c2 = used - live0;
      const uint32_t *p = data0 + (c2 >> 5);
      const unsigned rem = (unsigned)(c2 & 31u);
      if (rem) { bs->buff = ((uint64_t)ntohl(*p) << 32) << rem; bs->live = 32u - rem; p++; }
      else { bs->buff = 0; bs->live = 0; }
      bs->data = p;
    }
  };
  if (rec.err || rec.bit_used < 32u + before) {
    if (rec.bit_used >= 32u + before && rec.bit_used <= r.avail) stand_behind(rec.bit_used - 32u - before);
    else { bs->live = 0; bs->buff = 0; bs->data = bs->limit; }
    return wd_error(rec.err, rec.nblock);
  }
  stand_behind(rec.bit_used - 32u - before);
  ds->rand = rec.randomised != 0u;
  ds->bwt_idx = rec.orig_ptr;
  ds->block_size = rec.nblock;
  st->crc = rec.computed_crc;
  st->pos = 0;
  std::vector<uint8_t>().swap(st->bits);
  return WD_OK;
}

extern "C" void lbzamd_decode(struct decoder_state *ds) { (void)ds; }     /* the inverse BWT ran on the device with the codes */

extern "C" int lbzamd_emit(struct decoder_state *ds, void *buf, size_t *buf_sz)
{
  wd_state *st = reinterpret_cast<wd_state *>(ds->internal_state);
  if (!st || !buf || !buf_sz) { g_err = "emit(): bad decoder state"; die("emit"); }
  const size_t left = st->out.size() - st->pos;
  const size_t n = left < *buf_sz ? left : *buf_sz;
  if (n) memcpy(buf, st->out.data() + st->pos, n);
  st->pos += n;
  *buf_sz -= n;
  if (st->pos < st->out.size()) return WD_MORE;
  if (st->runlen) return WD_ERR_RUNLEN;
  ds->crc = st->crc;                                           /* decode.c:1141 */
  return WD_OK;
}

/* the reference's own symbol names (decode.h:77-81) */
extern "C" void decoder_init(struct decoder_state *ds) { lbzamd_decoder_init(ds); }
extern "C" void decoder_free(struct decoder_state *ds) { lbzamd_decoder_free(ds); }
extern "C" int retrieve(struct decoder_state *ds, struct bitstream *bs) { return lbzamd_retrieve(ds, bs); }
extern "C" void decode(struct decoder_state *ds) { lbzamd_decode(ds); }
extern "C" int emit(struct decoder_state *ds, void *buf, size_t *buf_sz) { return lbzamd_emit(ds, buf, buf_sz); }
