/*
 * lbz_dev.h -- wavefront / workgroup primitives for the gfx950 kernels.
 *
 * CDNA4 model used throughout: 64-lane wavefronts, LBZ_NW waves per workgroup, one
 * workgroup per bzip2 block; cross-lane traffic by shuffles/ballots, cross-wave traffic
 * by LDS; every scan below is "wave scan -> LBZ_NW partials in LDS -> wave 0 scans them".
 */
#ifndef LBZ_DEV_H
#define LBZ_DEV_H

#include <hip/hip_runtime.h>

#include "lbz_common.h"
#include <lbz_asm.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u32 wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ u64 lanes_below() { return (1ull << lane_id()) - 1ull; }

/* Lanes of one wave run in lockstep on the hardware; this only pins the compiler's
 * ordering of LDS accesses around a wave-private read-modify-write.                  */
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

/* Workgroup barrier that orders LDS traffic only: outstanding global stores keep flying
 * (a plain __syncthreads() drains vmcnt too, which exposes the full store latency at every
 * tile boundary of a streaming pass).  0xC07F = s_waitcnt lgkmcnt(0) on gfx9.             */
__device__ __forceinline__ void wg_lds_barrier()
{
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
}

template <class T> __device__ __forceinline__ T wave_incl_add(T v)
{
  const u32 l = lane_id();
#pragma unroll
  for (u32 d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d);
    if (l >= d) v += o;
  }
  return v;
}

template <class T> __device__ __forceinline__ T wave_incl_max(T v)
{
  const u32 l = lane_id();
#pragma unroll
  for (u32 d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d);
    if (l >= d && o > v) v = o;
  }
  return v;
}

template <class T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
  for (u32 d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, (int)d);
  return v;
}

template <class T> __device__ __forceinline__ T wave_min(T v)
{
#pragma unroll
  for (u32 d = 32; d >= 1; d >>= 1) { T o = __shfl_xor(v, (int)d); if (o < v) v = o; }
  return v;
}

template <class T> __device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
  for (u32 d = 32; d >= 1; d >>= 1) { T o = __shfl_xor(v, (int)d); if (o > v) v = o; }
  return v;
}

#ifndef LBZ_EMULATED
/* 32-bit unsigned scans and reductions on the DPP network instead of ds_bpermute: the kernels that scan the
 * most are the ones bound by LDS, and a __shfl_* is an LDS-crossbar operation.  row_shr 1/2/4/8 inside the rows
 * of 16, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; lanes without a source take the
 * identity.  Every one of these is a single v_<op>_dpp.                                                      */
#define LBZ_DPP(ident, v, ctrl, rows) ((u32)__builtin_amdgcn_update_dpp((int)(ident), (int)(v), (ctrl), (rows), 0xf, false))
#define LBZ_DPP_SCAN(OP, ident)                                          \
  v = OP(v, LBZ_DPP(ident, v, 0x111, 0xf));                              \
  v = OP(v, LBZ_DPP(ident, v, 0x112, 0xf));                              \
  v = OP(v, LBZ_DPP(ident, v, 0x114, 0xf));                              \
  v = OP(v, LBZ_DPP(ident, v, 0x118, 0xf));                              \
  v = OP(v, LBZ_DPP(ident, v, 0x142, 0xa));                              \
  v = OP(v, LBZ_DPP(ident, v, 0x143, 0xc));
__device__ __forceinline__ u32 dpp_add(u32 a, u32 b) { return a + b; }
__device__ __forceinline__ u32 dpp_max(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 dpp_min(u32 a, u32 b) { return a < b ? a : b; }
template <> __device__ __forceinline__ u32 wave_incl_add<u32>(u32 v) { LBZ_DPP_SCAN(dpp_add, 0u) return v; }
template <> __device__ __forceinline__ u32 wave_incl_max<u32>(u32 v) { LBZ_DPP_SCAN(dpp_max, 0u) return v; }
template <> __device__ __forceinline__ u32 wave_sum<u32>(u32 v) { LBZ_DPP_SCAN(dpp_add, 0u) return (u32)__builtin_amdgcn_readlane((int)v, 63); }
template <> __device__ __forceinline__ u32 wave_max<u32>(u32 v) { LBZ_DPP_SCAN(dpp_max, 0u) return (u32)__builtin_amdgcn_readlane((int)v, 63); }
template <> __device__ __forceinline__ u32 wave_min<u32>(u32 v) { LBZ_DPP_SCAN(dpp_min, 0xFFFFFFFFu) return (u32)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ u32 dpp_or(u32 a, u32 b) { return a | b; }
__device__ __forceinline__ u32 dpp_and(u32 a, u32 b) { return a & b; }
__device__ __forceinline__ u32 wave_or(u32 v) { LBZ_DPP_SCAN(dpp_or, 0u) return (u32)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ u32 wave_and(u32 v) { LBZ_DPP_SCAN(dpp_and, 0xFFFFFFFFu) return (u32)__builtin_amdgcn_readlane((int)v, 63); }
#else
__device__ __forceinline__ u32 wave_or(u32 v) { for (u32 d = 32; d >= 1; d >>= 1) v |= __shfl_xor(v, (int)d); return v; }
__device__ __forceinline__ u32 wave_and(u32 v) { for (u32 d = 32; d >= 1; d >>= 1) v &= __shfl_xor(v, (int)d); return v; }
#endif
/* Lanes of the wave holding the same 8-bit digit ("match any"): 8 ballots; each ballot is
 * folded in with an xnor against the lane's own sign-extended bit.                        */
__device__ __forceinline__ u64 match_digit(u32 d, bool ok)
{
  const u64 act = __ballot(ok);
  u32 lo = (u32)act, hi = (u32)(act >> 32);
#pragma unroll
  for (u32 b = 0; b < 8u; b++) {
    const int bm = -(int)((d >> b) & 1u);               /* 0 or ~0 */
    const u64 bal = __ballot(bm != 0);
    lo &= ~((u32)bal ^ (u32)bm);
    hi &= ~((u32)(bal >> 32) ^ (u32)bm);
  }
  return ((u64)hi << 32) | lo;
}

/* inclusive OR-scan over the lanes of a wave (DPP on the device) and its 64-bit exclusive form: lane l gets the OR of
   the values of lanes < l */
#ifndef LBZ_EMULATED
__device__ __forceinline__ u32 wave_incl_or(u32 v) { LBZ_DPP_SCAN(dpp_or, 0u) return v; }
#else
__device__ __forceinline__ u32 wave_incl_or(u32 v)
{
  const u32 l = lane_id();
  for (u32 d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(v, d); if (l >= d) v |= o; }
  return v;
}
#endif
__device__ __forceinline__ u64 wave_excl_or64(u64 v, u64 *total = nullptr)
{
  const u32 lo = wave_incl_or((u32)v), hi = wave_incl_or((u32)(v >> 32));
  if (total) *total = (u64)(u32)__builtin_amdgcn_readlane((int)lo, 63) | ((u64)(u32)__builtin_amdgcn_readlane((int)hi, 63) << 32);
  const u32 elo = (u32)wave_shr1((int)lo), ehi = (u32)wave_shr1((int)hi);       /* lane 0 keeps its own value: cleared below */
  return lane_id() == 0u ? 0ull : ((u64)elo | ((u64)ehi << 32));
}

/* OR and AND of a 64-bit value over the wave, in every lane */
__device__ __forceinline__ void wave_or_and64(u64 *vo, u64 *va)
{
  *vo = (u64)wave_or((u32)*vo) | (u64)wave_or((u32)(*vo >> 32)) << 32;
  *va = (u64)wave_and((u32)*va) | (u64)wave_and((u32)(*va >> 32)) << 32;
}
/* lane l takes lane l - 1's / l + 1's value (the end lane keeps its own): one DPP move, no LDS */
__device__ __forceinline__ u32 lane_from_below(u32 v) { return (u32)wave_shr1((int)v); }
__device__ __forceinline__ u32 lane_from_above(u32 v) { return (u32)wave_shl1((int)v); }

/* LDS scratch for the workgroup scans/reductions: one object, reused everywhere. */
struct wg_scratch {
  u32 a[LBZ_NW + 1];
  u32 b[LBZ_NW + 1];
};

/* Exclusive add-scan over the workgroup.  Returns this thread's exclusive prefix,
 * *total = sum over all threads.  Three barriers; sc may be reused right after.       */
__device__ __forceinline__ u32 wg_excl_add(u32 v, u32 *total, wg_scratch *sc)
{
  const u32 l = lane_id(), w = wave_id();
  u32 inc = wave_incl_add(v);
  if (l == 63) sc->a[w] = inc;
  __syncthreads();
  if (w == 0) {
    u32 p = l < LBZ_NW ? sc->a[l] : 0u;
    u32 pi = wave_incl_add(p);
    if (l < LBZ_NW) sc->a[l] = pi - p;
    if (l == LBZ_NW - 1) sc->a[LBZ_NW] = pi;
  }
  __syncthreads();
  u32 r = sc->a[w] + inc - v;
  *total = sc->a[LBZ_NW];
  __syncthreads();
  return r;
}

/* Exclusive max-scan (identity 0) and exclusive add-scan in one pass. */
__device__ __forceinline__ void wg_excl_max_add(u32 vmax, u32 vadd, u32 *emax, u32 *eadd,
                                                 u32 *tmax, u32 *tadd, wg_scratch *sc)
{
  const u32 l = lane_id(), w = wave_id();
  u32 im = wave_incl_max(vmax);
  u32 ia = wave_incl_add(vadd);
  if (l == 63) { sc->a[w] = im; sc->b[w] = ia; }
  __syncthreads();
  if (w == 0) {
    u32 pm = l < LBZ_NW ? sc->a[l] : 0u;
    u32 pa = l < LBZ_NW ? sc->b[l] : 0u;
    u32 qm = wave_incl_max(pm);
    u32 qa = wave_incl_add(pa);
    u32 qm_ex = lane_from_below(qm);
    if (l == 0) qm_ex = 0u;
    if (l < LBZ_NW) { sc->a[l] = qm_ex; sc->b[l] = qa - pa; }
    if (l == LBZ_NW - 1) { sc->a[LBZ_NW] = qm; sc->b[LBZ_NW] = qa; }
  }
  __syncthreads();
  u32 pm_ex = lane_from_below(im);
  if (l == 0) pm_ex = 0u;
  u32 bm = sc->a[w];
  *emax = bm > pm_ex ? bm : pm_ex;
  *eadd = sc->b[w] + ia - vadd;
  *tmax = sc->a[LBZ_NW];
  *tadd = sc->b[LBZ_NW];
  __syncthreads();
}

__device__ __forceinline__ u32 wg_min(u32 v, wg_scratch *sc)
{
  const u32 l = lane_id(), w = wave_id();
  v = wave_min(v);
  if (l == 0) sc->a[w] = v;
  __syncthreads();
  u32 r = sc->a[0];
#pragma unroll
  for (u32 i = 1; i < LBZ_NW; i++) { u32 o = sc->a[i]; if (o < r) r = o; }
  __syncthreads();
  return r;
}

__device__ __forceinline__ u32 wg_max(u32 v, wg_scratch *sc)
{
  const u32 l = lane_id(), w = wave_id();
  v = wave_max(v);
  if (l == 0) sc->a[w] = v;
  __syncthreads();
  u32 r = sc->a[0];
#pragma unroll
  for (u32 i = 1; i < LBZ_NW; i++) { u32 o = sc->a[i]; if (o > r) r = o; }
  __syncthreads();
  return r;
}

/* OR and AND of a 64-bit value over the workgroup (which key bytes vary at all?) */
__device__ __forceinline__ void wg_or_and64(u64 v_or, u64 v_and, u64 *r_or, u64 *r_and, wg_scratch *sc)
{
  const u32 l = lane_id(), w = wave_id();
  wave_or_and64(&v_or, &v_and);
  if (l == 0) { sc->a[w] = (u32)v_or; sc->b[w] = (u32)(v_or >> 32); }
  __syncthreads();
  u64 o = 0;
#pragma unroll
  for (u32 i = 0; i < LBZ_NW; i++) o |= (u64)sc->a[i] | ((u64)sc->b[i] << 32);
  __syncthreads();
  if (l == 0) { sc->a[w] = (u32)v_and; sc->b[w] = (u32)(v_and >> 32); }
  __syncthreads();
  u64 a = ~0ull;
#pragma unroll
  for (u32 i = 0; i < LBZ_NW; i++) a &= (u64)sc->a[i] | ((u64)sc->b[i] << 32);
  __syncthreads();
  *r_or = o;
  *r_and = a;
}

__device__ __forceinline__ u32 wg_sum(u32 v, wg_scratch *sc)
{
  const u32 l = lane_id(), w = wave_id();
  v = wave_sum(v);
  if (l == 0) sc->a[w] = v;
  __syncthreads();
  u32 r = 0;
#pragma unroll
  for (u32 i = 0; i < LBZ_NW; i++) r += sc->a[i];
  __syncthreads();
  return r;
}

#endif
