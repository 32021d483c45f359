/* lbz_asm.h -- the few gfx950 instructions hipcc has no builtin for. */
#ifndef LBZ_ASM_H
#define LBZ_ASM_H

/* v_writelane_b32: lane `lane` of the result takes the wave-uniform `val`, the other lanes keep
 * `old`.  Both scalars travel in SGPRs, so a serial per-element loop stays off the vector ALU. */
__device__ __forceinline__ int lane_write(int old, int val, int lane)
{
  /* gfx9 reads one scalar operand per vector instruction: the lane select goes through M0 */
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0"
               : "+v"(old) : "s"(__builtin_amdgcn_readfirstlane(val)), "s"(__builtin_amdgcn_readfirstlane(lane)) : "m0");
  return old;
}

/* One claim per wave from an LDS ticket counter (all 64 lanes active): returns 0, 1, 2, ... in
 * claim order, wave-uniform.  Every lane adds one -- the compiler folds that into a single
 * ds_add_rtn of 64 by one lane -- because a claim written as `if (lane == 0) atomicAdd` inside a
 * loop gets lane 0 peeled onto its own path and the wave never reconverges for the broadcast. */
__device__ __forceinline__ unsigned wave_claim(unsigned *tickets)
{
  const unsigned r = atomicAdd(tickets, 1u);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r) >> 6;
}

/* MTF ranks of consecutive run heads whose symbol slot is < 64 (k_mtf.hip, mtf_ranks): the loop
 * the compiler would not keep tight.  `heads` = ballot of the strip's run heads still to do,
 * `c` = slot of every lane's symbol, `L0..` = "last seen at" registers (register 0 = slots
 * 0..63), `sb0` = position of lane 0.  Per head: rank = #{slots last seen after this one},
 * written to its lane of `rank`; the head's position becomes its slot's entry in L0.  Stops at
 * the first head with a slot >= 64 (left in `heads`) or when none is left.
 * Wait states (none are inserted inside an asm block): 4 between the v_readlane that produces a
 * lane select and the lane access that uses it, 2 before a VALU reads an SGPR a VALU wrote,
 * 1 after s_mov m0.                                                                        */
#define LBZ_MTF_FAST_HEAD                                                                       \
  "s_cmp_eq_u64 %[h], 0\n\t"                                                                    \
  "s_cbranch_scc1 9f\n"                                                                         \
  "1:\n\t"                                                                                      \
  "s_ff1_i32_b64 %[l], %[h]\n\t"                                                                \
  "v_readlane_b32 %[s], %[c], %[l]\n\t"                                                         \
  "s_mov_b32 m0, %[l]\n\t"                                                                      \
  "s_add_i32 %[np], %[sb0], %[l]\n\t"                                                           \
  "s_cmp_gt_u32 %[s], 63\n\t"                                                                   \
  "s_cbranch_scc1 9f\n\t"                                                                       \
  "s_bitset0_b64 %[h], %[l]\n\t"                                                                \
  "v_readlane_b32 %[pv], %[L0], %[s]\n\t"                                                       \
  "s_nop 1\n\t"                                                                                 \
  "v_cmp_lt_i32_e32 vcc, %[pv], %[L0]\n\t"                                                      \
  "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
#define LBZ_MTF_FAST_MORE(R)                                                                    \
  "v_cmp_lt_i32_e32 vcc, %[pv], %[" R "]\n\t"                                                   \
  "s_bcnt1_i32_b64 %[t], vcc\n\t"                                                               \
  "s_add_i32 %[cnt], %[cnt], %[t]\n\t"
#define LBZ_MTF_FAST_TAIL                                                                       \
  "v_writelane_b32 %[rank], %[cnt], m0\n\t"                                                     \
  "s_mov_b32 m0, %[s]\n\t"                                                                      \
  "s_nop 0\n\t"                                                                                 \
  "v_writelane_b32 %[L0], %[np], m0\n\t"                                                        \
  "s_cmp_lg_u64 %[h], 0\n\t"                                                                    \
  "s_cbranch_scc1 1b\n"                                                                         \
  "9:\n\t"

__device__ __forceinline__ void mtf_fast_heads(unsigned long long &heads, int &rank, int (&L)[2], int c, int sb0)
{
  int l, s, np, pv, cnt, t;
  asm volatile(LBZ_MTF_FAST_HEAD LBZ_MTF_FAST_MORE("L1") LBZ_MTF_FAST_TAIL
               : [h] "+s"(heads), [rank] "+v"(rank), [L0] "+v"(L[0]),
                 [l] "=&s"(l), [s] "=&s"(s), [np] "=&s"(np), [pv] "=&s"(pv), [cnt] "=&s"(cnt), [t] "=&s"(t)
               : [L1] "v"(L[1]), [c] "v"(c), [sb0] "s"(sb0)
               : "vcc", "scc", "m0");
}
__device__ __forceinline__ void mtf_fast_heads(unsigned long long &heads, int &rank, int (&L)[4], int c, int sb0)
{
  int l, s, np, pv, cnt, t;
  asm volatile(LBZ_MTF_FAST_HEAD LBZ_MTF_FAST_MORE("L1") LBZ_MTF_FAST_MORE("L2") LBZ_MTF_FAST_MORE("L3") LBZ_MTF_FAST_TAIL
               : [h] "+s"(heads), [rank] "+v"(rank), [L0] "+v"(L[0]),
                 [l] "=&s"(l), [s] "=&s"(s), [np] "=&s"(np), [pv] "=&s"(pv), [cnt] "=&s"(cnt), [t] "=&s"(t)
               : [L1] "v"(L[1]), [L2] "v"(L[2]), [L3] "v"(L[3]), [c] "v"(c), [sb0] "s"(sb0)
               : "vcc", "scc", "m0");
}
__device__ __forceinline__ void mtf_fast_heads(unsigned long long &, int &, int (&)[1], int, int) {}   /* one register: no split */

#endif
