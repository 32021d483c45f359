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

/* wave_shr:1 -- lane l of the result is lane l - 1 of v (lane 0: v's own lane 0) */
__device__ __forceinline__ int wave_shr1(int v)
{
  return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
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

/* MTF ranks of the run heads in `f` (a set of lanes of one 64-position strip, all with symbol
 * slots < 64; k_mtf.hip, mtf_ranks), in lane order: the loop the compiler would not keep tight.
 * `c` = slot of every lane's symbol, `L0..` = "last seen at" registers (register 0 = slots
 * 0..63), `sb0` = position of lane 0.  Per head: rank = #{slots last seen after this one},
 * written to its lane of `rank`; the head's position becomes its slot's entry in L0.
 *
 * The loop is bound by the latency of its VALU->SGPR->VALU hops, not by issue slots, so it is
 * software-pipelined: while the compares of head i drain into their popcounts, the symbol of
 * head i+1 is already being fetched.  The only serial chain left per head is
 * readlane(pv) -> compares -> writelane(L0).
 * Wait states (none are inserted inside an asm block): 4 between the v_readlane that produces a
 * lane select and the lane access that uses it, 2 before a VALU reads an SGPR a VALU wrote,
 * 1 after s_mov m0.                                                                        */
#define LBZ_MTF_PIPE_HEAD                                                                       \
  "s_cmp_eq_u64 %[f], 0\n\t"                                                                    \
  "s_cbranch_scc1 9f\n\t"                                                                       \
  "s_ff1_i32_b64 %[l], %[f]\n\t"                                                                \
  "s_bitset0_b64 %[f], %[l]\n\t"                                                                \
  "v_readlane_b32 %[s], %[c], %[l]\n\t"                                                         \
  "s_add_i32 %[np], %[sb0], %[l]\n\t"                                                           \
  "s_nop 3\n"                                                                                   \
  "1:\n\t"                                                                                      \
  "v_readlane_b32 %[pv], %[L0], %[s]\n\t"                                                       \
  "s_mov_b32 m0, %[s]\n\t"                                                                      \
  "s_ff1_i32_b64 %[l2], %[f]\n\t"                                                               \
  "s_cmp_eq_u64 %[f], 0\n\t"                                                                    \
  "v_cmp_lt_i32_e64 %[sA], %[pv], %[L0]\n\t"
#define LBZ_MTF_PIPE_CMP(M, R) "v_cmp_lt_i32_e64 %[" M "], %[pv], %[" R "]\n\t"
#define LBZ_MTF_PIPE_MID                                                                        \
  "v_writelane_b32 %[L0], %[np], m0\n\t"                                                        \
  "s_cselect_b32 %[l2], 0, %[l2]\n\t"                                                           \
  "v_readlane_b32 %[s2], %[c], %[l2]\n\t"                                                       \
  "s_mov_b32 m0, %[l]\n\t"                                                                      \
  "s_add_i32 %[np2], %[sb0], %[l2]\n\t"                                                         \
  "s_bcnt1_i32_b64 %[cnt], %[sA]\n\t"
#define LBZ_MTF_PIPE_CNT(M)                                                                     \
  "s_bcnt1_i32_b64 %[t], %[" M "]\n\t"                                                          \
  "s_add_i32 %[cnt], %[cnt], %[t]\n\t"
#define LBZ_MTF_PIPE_TAIL                                                                       \
  "v_writelane_b32 %[rank], %[cnt], m0\n\t"                                                     \
  "s_cmp_eq_u64 %[f], 0\n\t"                                                                    \
  "s_cbranch_scc1 9f\n\t"                                                                       \
  "s_bitset0_b64 %[f], %[l2]\n\t"                                                               \
  "s_mov_b32 %[l], %[l2]\n\t"                                                                   \
  "s_mov_b32 %[s], %[s2]\n\t"                                                                   \
  "s_mov_b32 %[np], %[np2]\n\t"                                                                 \
  "s_branch 1b\n"                                                                               \
  "9:\n\t"

#define LBZ_MTF_PIPE_TEMPS                                                                      \
  [l] "=&s"(l), [s] "=&s"(s), [np] "=&s"(np), [pv] "=&s"(pv), [cnt] "=&s"(cnt), [t] "=&s"(t),   \
  [l2] "=&s"(l2), [s2] "=&s"(s2), [np2] "=&s"(np2)

__device__ __forceinline__ void mtf_fast_heads(unsigned long long f, int &rank, int (&L)[2], int c, int sb0)
{
  int l, s, np, pv, cnt, t, l2, s2, np2;
  unsigned long long sA, sB;
  asm volatile(LBZ_MTF_PIPE_HEAD LBZ_MTF_PIPE_CMP("sB", "L1") LBZ_MTF_PIPE_MID LBZ_MTF_PIPE_CNT("sB") LBZ_MTF_PIPE_TAIL
               : [f] "+s"(f), [rank] "+v"(rank), [L0] "+v"(L[0]), LBZ_MTF_PIPE_TEMPS, [sA] "=&s"(sA), [sB] "=&s"(sB)
               : [L1] "v"(L[1]), [c] "v"(c), [sb0] "s"(sb0)
               : "vcc", "scc", "m0");
}
__device__ __forceinline__ void mtf_fast_heads(unsigned long long f, int &rank, int (&L)[4], int c, int sb0)
{
  int l, s, np, pv, cnt, t, l2, s2, np2;
  unsigned long long sA, sB, sC, sD;
  asm volatile(LBZ_MTF_PIPE_HEAD LBZ_MTF_PIPE_CMP("sB", "L1") LBZ_MTF_PIPE_CMP("sC", "L2") LBZ_MTF_PIPE_CMP("sD", "L3")
               LBZ_MTF_PIPE_MID LBZ_MTF_PIPE_CNT("sB") LBZ_MTF_PIPE_CNT("sC") LBZ_MTF_PIPE_CNT("sD") LBZ_MTF_PIPE_TAIL
               : [f] "+s"(f), [rank] "+v"(rank), [L0] "+v"(L[0]), LBZ_MTF_PIPE_TEMPS,
                 [sA] "=&s"(sA), [sB] "=&s"(sB), [sC] "=&s"(sC), [sD] "=&s"(sD)
               : [L1] "v"(L[1]), [L2] "v"(L[2]), [L3] "v"(L[3]), [c] "v"(c), [sb0] "s"(sb0)
               : "vcc", "scc", "m0");
}
__device__ __forceinline__ void mtf_fast_heads(unsigned long long, int &, int (&)[1], int, int) {}   /* one register: no split */

#endif
