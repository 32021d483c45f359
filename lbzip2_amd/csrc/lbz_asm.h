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

#endif
