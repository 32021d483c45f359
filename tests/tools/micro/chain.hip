// Ad-hoc microbenchmark (tuning aid, not part of the product): cost per step of the serial chains the decoder's
// prefix-code stage is made of, one wave, nothing else on the SIMD.  Prints cycles (100 MHz ticks * 24 at 2.4 GHz
// is only nominal; ns are what counts).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

__global__ void k_chain(u32 *out, u64 *ticks, u32 iters, u32 mode)
{
  __shared__ u32 tab[4096];
  const u32 lane = threadIdx.x;
  for (u32 i = lane; i < 4096u; i += 64u) tab[i] = (i * 2654435761u >> 7) & 4095u;
  __syncthreads();
  u32 x = rfl(out[0]) & 4095u, acc = 0;
  int L0 = (int)lane;
  const u64 t0 = wall_clock64();
  if (mode == 0u) {                    // LDS lookup chain through readfirstlane (uniform address)
    for (u32 i = 0; i < iters; i++) x = rfl(tab[x]);
  } else if (mode == 1u) {             // per-lane LDS chain, no readfirstlane
    u32 y = x + lane;
    for (u32 i = 0; i < iters; i++) y = tab[y & 4095u];
    x = y;
  } else if (mode == 2u) {             // scalar ALU chain
    for (u32 i = 0; i < iters; i++) { x = x * 5u + 1u; x ^= x >> 3; x += 7u; x ^= x << 2; }
  } else if (mode == 3u) {             // readlane <-> scalar hops
    for (u32 i = 0; i < iters; i++) { x = (u32)__builtin_amdgcn_readlane(L0, (int)(x & 63u)); x = x * 5u + 1u; L0 += 1; }
  } else if (mode == 4u) {             // lookup + shift + front move + byte store: the fast path's shape
    u64 buf = 0x0123456789abcdefull * (x + 1u);
    u32 n = 0;
    for (u32 i = 0; i < iters; i++) {
      const u32 e = rfl(tab[(u32)(buf >> 52)]);
      const u32 l = (e & 7u) + 1u, nn = (e >> 3) & 63u;
      buf = (buf << l) | (buf >> (64u - l));
      const u32 m = (u32)__builtin_amdgcn_readlane(L0, (int)nn);
      const int sh = __builtin_amdgcn_update_dpp(L0, L0, 0x138, 0xf, 0xf, false);
      L0 = lane == 0u ? (int)m : (lane <= nn ? sh : L0);
      if (lane == 0u) reinterpret_cast<unsigned char *>(out + 1024)[n & 0xFFFFu] = (unsigned char)m;
      n++;
    }
    x = (u32)buf;
  } else if (mode == 5u) {             // same without the store
    u64 buf = 0x0123456789abcdefull * (x + 1u);
    for (u32 i = 0; i < iters; i++) {
      const u32 e = rfl(tab[(u32)(buf >> 52)]);
      const u32 l = (e & 7u) + 1u, nn = (e >> 3) & 63u;
      buf = (buf << l) | (buf >> (64u - l));
      const u32 m = (u32)__builtin_amdgcn_readlane(L0, (int)nn);
      const int sh = __builtin_amdgcn_update_dpp(L0, L0, 0x138, 0xf, 0xf, false);
      L0 = lane == 0u ? (int)m : (lane <= nn ? sh : L0);
    }
    x = (u32)buf;
  } else if (mode == 6u) {             // lookup + shift only
    u64 buf = 0x0123456789abcdefull * (x + 1u);
    for (u32 i = 0; i < iters; i++) {
      const u32 e = rfl(tab[(u32)(buf >> 52)]);
      const u32 l = (e & 7u) + 1u;
      buf = (buf << l) | (buf >> (64u - l));
      acc += e;
    }
    x = (u32)buf;
  }
  const u64 t1 = wall_clock64();
  if (lane == 0u) { out[blockIdx.x * 2u + 2u] = x + acc + (u32)L0; ticks[blockIdx.x] = t1 - t0; }
}

int main()
{
  u32 *out; u64 *ticks;
  hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20);
  hipMalloc(&ticks, 4096 * 8);
  const u32 iters = 200000;
  const char *names[] = { "lds->readfirstlane chain", "lds per-lane chain", "salu chain (8 ops)", "readlane<->salu hop", "fast path shape + store", "fast path shape", "lookup + shift" };
  for (u32 waves : { 1u, 1024u, 4096u })
    for (u32 mode = 0; mode < 7; mode++) {
      hipLaunchKernelGGL(k_chain, dim3(waves), dim3(64), 0, 0, out, ticks, iters, mode);
      hipDeviceSynchronize();
      u64 t[4096]; hipMemcpy(t, ticks, waves * 8, hipMemcpyDeviceToHost);
      u64 mx = 0; for (u32 i = 0; i < waves; i++) mx = t[i] > mx ? t[i] : mx;
      printf("waves %4u  %-28s %7.1f ns/step\n", waves, names[mode], mx * 10.0 / iters);
    }
  return 0;
}
