/* TEST INFRASTRUCTURE: emulator counterpart of lbzip2_amd/csrc/lbz_asm.h (found first through
 * -Itests/emu).  val and lane are wave-uniform. */
#ifndef LBZ_ASM_H
#define LBZ_ASM_H
static inline int lane_write(int old, int val, int lane) { return (int)(threadIdx.x & 63) == lane ? val : old; }
static inline int wave_shr1(int v) { const int o = __shfl_up(v, 1u); return (threadIdx.x & 63) == 0 ? v : o; }
static inline int wave_shl1(int v) { const int o = __shfl_down(v, 1u); return (threadIdx.x & 63) == 63 ? v : o; }
/* C statement of lbz_asm.h's huff_fast (same contract; lut is the table itself here, not its LDS address) */
static inline void huff_fast(unsigned long long &buf, unsigned &live, unsigned &dwl, unsigned cur, unsigned &k,
                             unsigned &n, unsigned &es, unsigned &N, int &L0, int &L1, int &L2, int &L3, const unsigned short *lut,
                             unsigned char *tt8, unsigned maxn, unsigned lane)
{
  for (;;) {
    /* a strip: lane j looks up the 10 bits that start at bit j of the buffer */
    if (live <= 32u) {
      const unsigned i = dwl & 63u;
      if (i == 63u) return;
      const unsigned v = (unsigned)__builtin_amdgcn_readlane((int)cur, (int)i);
      buf |= (unsigned long long)v << (32u - live);
      live += 32u; dwl++;
    }
    const int ve = (int)lut[(unsigned)((buf << lane) >> 54)];
    unsigned off = 0;
    const unsigned lim = live - 10u < 53u ? live - 10u : 53u;      /* a strip ends before bit 64 (shift counts 0..63) */
    for (;;) {
      if (k >= 50u) { buf <<= off; live -= off; return; }
      if (off > lim) break;
      const unsigned e = (unsigned)__builtin_amdgcn_readlane(ve, (int)off);
      if (e == 0u) { buf <<= off; live -= off; return; }
      const unsigned l = e & 31u, sym = e >> 5;
      if (sym <= 1u) {
        if (N >= 21u) { buf <<= off; live -= off; return; }
        es += (sym + 1u) << N; N++;
      } else {
        const unsigned nn = sym - 1u;
        if (es) {
          if (es > 64u || n + es > maxn) { buf <<= off; live -= off; return; }
          const unsigned uc = (unsigned)__builtin_amdgcn_readlane(L0, 0);
          if (lane < es) tt8[n + lane] = (unsigned char)uc;
          n += es; es = 0; N = 0;
        }
        unsigned m;
        if (nn < 64u) {
          m = (unsigned)__builtin_amdgcn_readlane(L0, (int)nn);
          const int sh = wave_shr1(L0);
          L0 = lane == 0u ? (int)m : (lane <= nn ? sh : L0);
        } else {
          const unsigned q = nn >> 6, r = nn & 63u;
          const int c0 = __builtin_amdgcn_readlane(L0, 63), c1 = __builtin_amdgcn_readlane(L1, 63), c2 = __builtin_amdgcn_readlane(L2, 63);
          m = (unsigned)__builtin_amdgcn_readlane(q == 1u ? L1 : (q == 2u ? L2 : L3), (int)r);
          const int s0 = wave_shr1(L0), s1 = wave_shr1(L1), s2 = wave_shr1(L2), s3 = wave_shr1(L3);
          L0 = lane == 0u ? (int)m : s0;
          L1 = lane == 0u ? c0 : ((q > 1u || lane <= r) ? s1 : L1);
          if (q >= 2u) L2 = lane == 0u ? c1 : ((q > 2u || lane <= r) ? s2 : L2);
          if (q >= 3u) L3 = lane == 0u ? c2 : (lane <= r ? s3 : L3);
        }
        if (lane == 0u) tt8[n] = (unsigned char)m;
        n++;
      }
      off += l; k++;
    }
    buf <<= off; live -= off;
  }
}
/* lanes are fibers here and their atomics interleave with other waves': lane 0 takes the whole
 * wave's 64 tickets at once */
static inline unsigned wave_claim(unsigned *tickets)
{
  unsigned r = 0;
  if ((threadIdx.x & 63) == 0) r = atomicAdd(tickets, 64u);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r) >> 6;
}
#endif
