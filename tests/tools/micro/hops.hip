// Ad-hoc microbenchmark (tuning aid, not part of the product): cost of one hop of the decoder's bit-chain walk
// (k_decode.hip, huff_walk) written in different ways; one wave, shader clock cycles per hop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;

__global__ void k_hops(u32 *out, u64 *cyc, u32 iters, u32 mode)
{
  __shared__ u32 tab[64];
  const u32 lane = threadIdx.x;
  // nx: a cycle through all 64 lanes so that the walk never stops: lane -> (lane + 7) & 63
  const u32 nx = (lane + 7u) & 63u, len = 7u;
  tab[lane] = nx;
  __syncthreads();
  u32 off = out[0] & 63u, acc = 0;
  u64 M = 0;
  const u64 t0 = clock64();
  if (mode == 0u) {            // readlane chain, lane select = previous readlane's result, bitset + s_nop 2 between (current huff_walk)
    for (u32 i = 0; i < iters; i += 4u) {
      u32 n1, n2, n3;
      asm volatile(
        "v_readlane_b32 %[n1], %[nx], %[off]\n\ts_bitset1_b64 %[M], %[off]\n\ts_nop 2\n\t"
        "v_readlane_b32 %[n2], %[nx], %[n1]\n\ts_bitset1_b64 %[M], %[n1]\n\ts_nop 2\n\t"
        "v_readlane_b32 %[n3], %[nx], %[n2]\n\ts_bitset1_b64 %[M], %[n2]\n\ts_nop 2\n\t"
        "v_readlane_b32 %[off], %[nx], %[n3]\n\ts_bitset1_b64 %[M], %[n3]\n\ts_nop 2\n\t"
        : [off] "+s"(off), [M] "+s"(M), [n1] "=&s"(n1), [n2] "=&s"(n2), [n3] "=&s"(n3) : [nx] "v"(nx));
    }
  } else if (mode == 1u) {     // the same without the bitsets (s_nop 3)
    for (u32 i = 0; i < iters; i += 4u) {
      u32 n1, n2, n3;
      asm volatile(
        "v_readlane_b32 %[n1], %[nx], %[off]\n\ts_nop 3\n\t"
        "v_readlane_b32 %[n2], %[nx], %[n1]\n\ts_nop 3\n\t"
        "v_readlane_b32 %[n3], %[nx], %[n2]\n\ts_nop 3\n\t"
        "v_readlane_b32 %[off], %[nx], %[n3]\n\ts_nop 3\n\t"
        : [off] "+s"(off), [n1] "=&s"(n1), [n2] "=&s"(n2), [n3] "=&s"(n3) : [nx] "v"(nx));
    }
  } else if (mode == 2u) {     // through the scalar ALU: readlane length, s_add, s_and
    for (u32 i = 0; i < iters; i += 4u) {
      u32 e;
      asm volatile(
        "v_readlane_b32 %[e], %[len], %[off]\n\ts_bitset1_b64 %[M], %[off]\n\ts_add_u32 %[off], %[off], %[e]\n\ts_and_b32 %[off], %[off], 63\n\t"
        "v_readlane_b32 %[e], %[len], %[off]\n\ts_bitset1_b64 %[M], %[off]\n\ts_add_u32 %[off], %[off], %[e]\n\ts_and_b32 %[off], %[off], 63\n\t"
        "v_readlane_b32 %[e], %[len], %[off]\n\ts_bitset1_b64 %[M], %[off]\n\ts_add_u32 %[off], %[off], %[e]\n\ts_and_b32 %[off], %[off], 63\n\t"
        "v_readlane_b32 %[e], %[len], %[off]\n\ts_bitset1_b64 %[M], %[off]\n\ts_add_u32 %[off], %[off], %[e]\n\ts_and_b32 %[off], %[off], 63\n\t"
        : [off] "+s"(off), [M] "+s"(M), [e] "=&s"(e) : [len] "v"(len));
    }
  } else if (mode == 3u) {     // through the scalar ALU with one op: readlane next, s_mov (copy) as the only SALU op
    for (u32 i = 0; i < iters; i += 4u) {
      u32 e;
      asm volatile(
        "v_readlane_b32 %[e], %[nx], %[off]\n\ts_bitset1_b64 %[M], %[off]\n\ts_mov_b32 %[off], %[e]\n\t"
        "v_readlane_b32 %[e], %[nx], %[off]\n\ts_bitset1_b64 %[M], %[off]\n\ts_mov_b32 %[off], %[e]\n\t"
        "v_readlane_b32 %[e], %[nx], %[off]\n\ts_bitset1_b64 %[M], %[off]\n\ts_mov_b32 %[off], %[e]\n\t"
        "v_readlane_b32 %[e], %[nx], %[off]\n\ts_bitset1_b64 %[M], %[off]\n\ts_mov_b32 %[off], %[e]\n\t"
        : [off] "+s"(off), [M] "+s"(M), [e] "=&s"(e) : [nx] "v"(nx));
    }
  } else if (mode == 4u) {     // vector only: ds_bpermute chain (every lane walks its own chain)
    u32 x = (off + lane) & 63u;
    for (u32 i = 0; i < iters; i++) x = (u32)__builtin_amdgcn_ds_bpermute((int)(x << 2), (int)nx);
    acc = x;
  } else if (mode == 5u) {     // LDS table, uniform address, readfirstlane
    for (u32 i = 0; i < iters; i++) off = (u32)__builtin_amdgcn_readfirstlane((int)tab[off]);
  } else if (mode == 6u) {     // m0 as the lane select (s_mov m0 + readlane with m0)
    for (u32 i = 0; i < iters; i += 4u) {
      u32 e;
      asm volatile(
        "s_mov_b32 m0, %[off]\n\ts_bitset1_b64 %[M], %[off]\n\tv_readlane_b32 %[off], %[nx], m0\n\t"
        "s_mov_b32 m0, %[off]\n\ts_bitset1_b64 %[M], %[off]\n\tv_readlane_b32 %[off], %[nx], m0\n\t"
        "s_mov_b32 m0, %[off]\n\ts_bitset1_b64 %[M], %[off]\n\tv_readlane_b32 %[off], %[nx], m0\n\t"
        "s_mov_b32 m0, %[off]\n\ts_bitset1_b64 %[M], %[off]\n\tv_readlane_b32 %[off], %[nx], m0\n\t"
        : [off] "+s"(off), [M] "+s"(M), [e] "=&s"(e) : [nx] "v"(nx) : "m0");
    }
  } else if (mode == 7u) {     // scalar only: the 64 next-offsets as bytes in 16 SGPRs, s_movrels + s_bfe
    // (table contents do not matter for the timing: use the loop-carried value itself)
    u32 t0r = 0x07060504u, idx, sh, w;
    for (u32 i = 0; i < iters; i += 2u) {
      asm volatile(
        "s_lshr_b32 %[idx], %[off], 2\n\ts_mov_b32 m0, %[idx]\n\ts_lshl_b32 %[sh], %[off], 3\n\ts_movrels_b32 %[w], %[t]\n\ts_lshr_b32 %[w], %[w], %[sh]\n\ts_and_b32 %[off], %[w], 3\n\t"
        "s_lshr_b32 %[idx], %[off], 2\n\ts_mov_b32 m0, %[idx]\n\ts_lshl_b32 %[sh], %[off], 3\n\ts_movrels_b32 %[w], %[t]\n\ts_lshr_b32 %[w], %[w], %[sh]\n\ts_and_b32 %[off], %[w], 3\n\t"
        : [off] "+s"(off), [idx] "=&s"(idx), [sh] "=&s"(sh), [w] "=&s"(w) : [t] "s"(t0r) : "m0");
    }
  }
  const u64 t1 = clock64();
  if (lane == 0u) { out[1] = off + acc + (u32)M + (u32)(M >> 32); cyc[0] = t1 - t0; }
}

int main()
{
  u32 *out; u64 *cyc;
  hipMalloc(&out, 4096); hipMemset(out, 0, 4096);
  hipMalloc(&cyc, 64);
  const u32 iters = 400000;
  const char *names[] = { "readlane->readlane, bitset + s_nop 2", "readlane->readlane, s_nop 3", "readlane -> s_add, s_and -> readlane", "readlane -> s_mov -> readlane",
                          "ds_bpermute chain (per lane)", "LDS table + readfirstlane", "s_mov m0 -> readlane m0", "scalar: s_movrels + shifts" };
  for (u32 mode = 0; mode < 8; mode++) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_hops, dim3(1), dim3(64), 0, 0, out, cyc, iters, mode); hipDeviceSynchronize(); }
    u64 c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s %6.1f cycles/hop\n", names[mode], (double)c / iters);
  }
  return 0;
}
