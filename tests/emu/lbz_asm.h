/* TEST INFRASTRUCTURE: emulator counterpart of lbzip2_amd/csrc/lbz_asm.h (found first through
 * -Itests/emu).  val and lane are wave-uniform. */
#ifndef LBZ_ASM_H
#define LBZ_ASM_H
static inline int lane_write(int old, int val, int lane) { return (int)(threadIdx.x & 63) == lane ? val : old; }
/* lanes are fibers here and their atomics interleave with other waves': lane 0 takes the whole
 * wave's 64 tickets at once */
static inline unsigned wave_claim(unsigned *tickets)
{
  unsigned r = 0;
  if ((threadIdx.x & 63) == 0) r = atomicAdd(tickets, 64u);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r) >> 6;
}
#endif
