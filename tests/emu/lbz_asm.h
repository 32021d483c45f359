/* TEST INFRASTRUCTURE: emulator counterpart of lbzip2_amd/csrc/lbz_asm.h (found first through
 * -Itests/emu).  val and lane are wave-uniform. */
#ifndef LBZ_ASM_H
#define LBZ_ASM_H
static inline int lane_write(int old, int val, int lane) { return (int)(threadIdx.x & 63) == lane ? val : old; }
static inline int wave_shr1(int v) { const int o = __shfl_up(v, 1u); return (threadIdx.x & 63) == 0 ? v : o; }
/* lanes are fibers here and their atomics interleave with other waves': lane 0 takes the whole
 * wave's 64 tickets at once */
static inline unsigned wave_claim(unsigned *tickets)
{
  unsigned r = 0;
  if ((threadIdx.x & 63) == 0) r = atomicAdd(tickets, 64u);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r) >> 6;
}
/* C statement of lbz_asm.h's mtf_fast_heads (same contract): every head in f has a slot < 64 */
template <int NQ>
static inline void mtf_fast_heads(unsigned long long f, int &rank, int (&L)[NQ], int c, int sb0)
{
  if (NQ == 1) return;
  while (f) {
    const int l = __builtin_ctzll(f);
    f &= f - 1ull;
    const int s = __builtin_amdgcn_readlane(c, l);
    const int pv = __builtin_amdgcn_readlane(L[0], s);
    int cnt = 0;
    for (int j = 0; j < NQ; j++) cnt += (int)__popcll(__ballot(L[j] > pv));
    rank = lane_write(rank, cnt, l);
    L[0] = lane_write(L[0], sb0 + l, s);
  }
}
#endif
