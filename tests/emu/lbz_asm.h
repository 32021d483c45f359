/* TEST INFRASTRUCTURE: emulator counterpart of lbzip2_amd/csrc/lbz_asm.h (found first through
 * -Itests/emu).  val and lane are wave-uniform. */
#ifndef LBZ_ASM_H
#define LBZ_ASM_H
static inline int lane_write(int old, int val, int lane) { return (int)(threadIdx.x & 63) == lane ? val : old; }
#endif
