/*
 * tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A CPU stand-in for <hip/hip_runtime.h> so that the kernel LOGIC in
 * lbzip2_amd/csrc/*.hip can be exercised in the GPU-less build container:
 * every GPU thread is a fiber, a workgroup's fibers are scheduled cooperatively on one
 * OS thread, __syncthreads()/wave collectives are fiber barriers, workgroups of a grid
 * run on a small pool of OS threads.  It models semantics (64-lane waves, LDS shared by
 * the workgroup, atomics), not performance, and is never part of the product: the
 * shipped library is built by hipcc for gfx950 only (lbzip2_amd/csrc/Makefile).
 * LBZ_EMU_CHECK_SITES=1|2: the wrappers below hand the source line of every wave collective to the runtime, which
 * reports (1) or aborts on (2) lanes of one wave that meet in a collective from different lines (emu_runtime.cpp).
 */
#ifndef LBZ_EMU_HIP_RUNTIME_H
#define LBZ_EMU_HIP_RUNTIME_H

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <unistd.h>

#define LBZ_EMULATED 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct ulonglong2 { unsigned long long x, y; };

namespace emu {
struct Idx { unsigned x, y, z; };
extern thread_local Idx threadIdx_, blockIdx_, blockDim_, gridDim_;
void launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem);
void sync_block();
void sync_wave();
extern thread_local const char *sitefile_;
extern thread_local int site_;   /* LBZ_EMU_CHECK_SITES: source line of the collective the fiber is about to enter */
unsigned long long wave_exchange(unsigned long long v, int src_lane);   /* value of src_lane (or own if dead) */
unsigned long long wave_ballot(int pred);
}
#define threadIdx (emu::threadIdx_)
#define blockIdx (emu::blockIdx_)
#define blockDim (emu::blockDim_)
#define gridDim (emu::gridDim_)
static const int warpSize = 64;

/* ---- launch ---- */
typedef struct ihipStream_t *hipStream_t;
typedef struct ihipEvent_t *hipEvent_t;
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(shmem))

/* ---- synchronisation & wave collectives ---- */
static inline void __syncthreads() { emu::sync_block(); }
static inline void __builtin_amdgcn_wave_barrier(int line = __builtin_LINE(), const char *file = __builtin_FILE()) { emu::site_ = line; emu::sitefile_ = file; emu::sync_wave(); }
static inline void __builtin_amdgcn_s_barrier() { emu::sync_block(); }
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline void __threadfence_block() {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_s_sleep(int) { usleep(50); }   /* a spinning workgroup waits for one that runs on another host thread */
static inline unsigned long long wall_clock64() { return 0; }
static inline unsigned long long clock64() { return 0; }

static inline unsigned long long __ballot(int pred, int line = __builtin_LINE(), const char *file = __builtin_FILE()) { emu::site_ = line; emu::sitefile_ = file; return emu::wave_ballot(pred); }
static inline int __any(int pred, int line = __builtin_LINE(), const char *file = __builtin_FILE()) { emu::site_ = line; emu::sitefile_ = file; return emu::wave_ballot(pred) != 0; }
static inline int __all(int pred, int line = __builtin_LINE(), const char *file = __builtin_FILE()) { emu::site_ = line; emu::sitefile_ = file; return emu::wave_ballot(!pred) == 0; }

template <class T> static inline T emu_xchg(T v, int src)
{
  static_assert(sizeof(T) <= 8, "shuffle of >8 bytes");
  unsigned long long raw = 0;
  memcpy(&raw, &v, sizeof(T));
  raw = emu::wave_exchange(raw, src);
  T r;
  memcpy(&r, &raw, sizeof(T));
  return r;
}
template <class T> static inline T __shfl(T v, int src, int width = 64, int line = __builtin_LINE(), const char *file = __builtin_FILE())
{
  emu::site_ = line; emu::sitefile_ = file;
  int lane = threadIdx.x & 63;
  int base = lane & ~(width - 1);
  return emu_xchg(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64, int line = __builtin_LINE(), const char *file = __builtin_FILE())
{
  emu::site_ = line; emu::sitefile_ = file;
  int lane = threadIdx.x & 63;
  int s = lane - (int)d;
  return emu_xchg(v, (s < (lane & ~(width - 1))) ? lane : s);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64, int line = __builtin_LINE(), const char *file = __builtin_FILE())
{
  emu::site_ = line; emu::sitefile_ = file;
  int lane = threadIdx.x & 63;
  int s = lane + (int)d;
  return emu_xchg(v, (s > (lane | (width - 1))) ? lane : s);
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64, int line = __builtin_LINE(), const char *file = __builtin_FILE())
{
  emu::site_ = line; emu::sitefile_ = file;
  int lane = threadIdx.x & 63;
  (void)width;
  return emu_xchg(v, lane ^ m);
}
static inline int __builtin_amdgcn_readfirstlane(int v, int line = __builtin_LINE(), const char *file = __builtin_FILE())
{
  emu::site_ = line; emu::sitefile_ = file;
  unsigned long long live = emu::wave_ballot(1);
  emu::site_ = line; emu::sitefile_ = file;        /* (other fibers have run since) */
  return emu_xchg(v, __builtin_ctzll(live));
}
static inline int __builtin_amdgcn_readlane(int v, int lane, int line = __builtin_LINE(), const char *file = __builtin_FILE()) { emu::site_ = line; emu::sitefile_ = file; return emu_xchg(v, lane); }
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned acc)
{ unsigned lane = threadIdx.x & 63; return acc + __builtin_popcount(m & (lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1))); }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned acc)
{ unsigned lane = threadIdx.x & 63; return acc + (lane > 32 ? __builtin_popcount(m & ((1u << (lane - 32)) - 1)) : 0); }

/* ---- bit ops ---- */
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline unsigned __brev(unsigned v)
{ unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }

/* ---- atomics (workgroups may run on different OS threads) ---- */
template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicSub(T *p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicCAS(T *p, T cmp, T v)
{ __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
template <class T> static inline T atomicMax(T *p, T v)
{ T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o; }
template <class T> static inline T atomicMin(T *p, T v)
{ T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o; }

/* ---- host API subset (synchronous) ---- */
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{ memset(p, 0, sizeof *p); strcpy(p->name, "emulated"); strcpy(p->gcnArchName, "emu"); p->multiProcessorCount = 8;
  p->totalGlobalMem = (size_t)16 << 30; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = n ? aligned_alloc(256, (n + 255) & ~(size_t)255) : nullptr; return (*p || !n) ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void **)p, n, f); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
/* every emulated allocation is host memory; LBZ_EMU_NO_HOSTPTR makes the runtime take its staging path instead */
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { if (getenv("LBZ_EMU_NO_HOSTPTR")) return hipErrorInvalidValue; *d = h; return hipSuccess; }
/* every host pointer counts as page-locked unless LBZ_EMU_PAGEABLE is set (the host-buffer calls then take the path of
   pageable input: a round's copy issued next to the launches of the round before it) */
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnregistered = 3 };
struct hipPointerAttribute_t { hipMemoryType type; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *)
{
  a->type = getenv("LBZ_EMU_PAGEABLE") ? hipMemoryTypeUnregistered : hipMemoryTypeHost;
  return hipSuccess;
}
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
/* LBZ_EMU_DEVICES=N: N fake devices (one address space), so that the per-device code paths of the host runtime run */
static inline int emu_ndev() { const char *e = getenv("LBZ_EMU_DEVICES"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n; }
static inline int &emu_curdev() { static thread_local int d = 0; return d; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= emu_ndev()) return hipErrorInvalidValue; emu_curdev() = d; return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = emu_curdev(); return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = emu_ndev(); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)8 << 30; *t = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
struct ihipEvent_t { double t; };
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new ihipEvent_t{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr)
{ e->t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipError(emu)"; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
#define hipHostMallocDefault 0
#define hipHostMallocPortable 1

#endif
