/* TEST INFRASTRUCTURE: emulator counterpart of lbzip2_amd/csrc/lbz_asm.h (found first through
 * -Itests/emu).  val and lane are wave-uniform. */
#ifndef LBZ_ASM_H
#define LBZ_ASM_H
static inline int lane_write(int old, int val, int lane) { return (int)(threadIdx.x & 63) == lane ? val : old; }
static inline int wave_shr1(int v) { const int o = __shfl_up(v, 1u); return (threadIdx.x & 63) == 0 ? v : o; }
static inline int wave_shl1(int v) { const int o = __shfl_down(v, 1u); return (threadIdx.x & 63) == 63 ? v : o; }
/* C statement of lbz_asm.h's huff_walk (same contract): the walks over the codes of even and of odd number */
static inline void huff_walk(unsigned nx, unsigned nx2, unsigned start, unsigned &off, unsigned long long &M)
{
  unsigned a = start, b = (unsigned)__builtin_amdgcn_readlane((int)nx, (int)start);
  M = 0;
  for (;;) {
    M |= 1ull << (a & 63u);
    M |= 1ull << (b & 63u);
    if (a & 64u) break;
    a = (unsigned)__builtin_amdgcn_readlane((int)nx2, (int)(a & 63u));
    b = (unsigned)__builtin_amdgcn_readlane((int)nx2, (int)(b & 63u));
  }
  off = a;
}
/* C statement of lbz_asm.h's huff_store (same contract) */
static inline void huff_store(unsigned short *sym16, unsigned at, unsigned e, unsigned long long M)
{
  const unsigned lane = threadIdx.x & 63;
  if ((M >> lane) & 1ull) sym16[at + (unsigned)__builtin_popcountll(M & ((1ull << lane) - 1ull))] = (unsigned short)(e >> 5);
}
/* C statement of lbz_asm.h's mtf_strip (same contract) */
static inline void mtf_strip(int &L0, int &L1, int &L2, int &L3, unsigned v, unsigned long long m, int &outv, unsigned lane)
{
  while (m) {
    const unsigned i = (unsigned)__builtin_ctzll(m);
    m &= m - 1ull;
    const unsigned nn = (unsigned)__builtin_amdgcn_readlane((int)v, (int)i) - 1u;
    unsigned x;
    if (nn < 64u) {
      x = (unsigned)__builtin_amdgcn_readlane(L0, (int)nn);
      const int sh = wave_shr1(L0);
      L0 = lane == 0u ? (int)x : (lane <= nn ? sh : L0);
    } else {
      const unsigned q = nn >> 6, r = nn & 63u;
      const int c0 = __builtin_amdgcn_readlane(L0, 63), c1 = __builtin_amdgcn_readlane(L1, 63), c2 = __builtin_amdgcn_readlane(L2, 63);
      x = (unsigned)__builtin_amdgcn_readlane(q == 1u ? L1 : (q == 2u ? L2 : L3), (int)r);
      const int s0 = wave_shr1(L0), s1 = wave_shr1(L1), s2 = wave_shr1(L2), s3 = wave_shr1(L3);
      L0 = lane == 0u ? (int)x : s0;
      L1 = lane == 0u ? c0 : ((q > 1u || lane <= r) ? s1 : L1);
      if (q >= 2u) L2 = lane == 0u ? c1 : ((q > 2u || lane <= r) ? s2 : L2);
      if (q >= 3u) L3 = lane == 0u ? c2 : (lane <= r ? s3 : L3);
    }
    outv = lane_write(outv, (int)(x + 2u), (int)i);
  }
}
/* lbz_asm.h's LDS counter: the waves of a workgroup are fibers of one host thread, and a wave collective is where
 * the scheduler moves on to the next -- so a waiting wave must look THROUGH one */
static inline void lds_publish(unsigned *p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static inline unsigned lds_observe(unsigned *p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)__atomic_load_n(p, __ATOMIC_ACQUIRE)); }
static inline void wave_pause() {}
/* lanes are fibers here and their atomics interleave with other waves': lane 0 takes the whole
 * wave's 64 tickets at once */
static inline unsigned wave_claim(unsigned *tickets)
{
  unsigned r = 0;
  if ((threadIdx.x & 63) == 0) r = atomicAdd(tickets, 64u);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r) >> 6;
}
/* lbz_asm.h's wave_reserve: lane 0 takes the places for the wave (see wave_claim) */
static inline unsigned wave_reserve(unsigned *counter, unsigned amount)
{
  unsigned r = 0;
  if ((threadIdx.x & 63) == 0) r = atomicAdd(counter, amount);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r);
}
struct __attribute__((packed, aligned(1))) lbz_text16 { unsigned long long a, b; };
struct __attribute__((packed, aligned(1))) lbz_text4 { unsigned a; };
static inline unsigned add_if_less2(unsigned acc, unsigned long long a0, unsigned long long a1, unsigned long long b) { return acc + (a0 < b ? 1u : 0u) + (a1 < b ? 1u : 0u); }
static inline unsigned long long ldg_u64(const unsigned long long *p) { return *p; }
static inline unsigned ldg_u32(const unsigned *p) { return *p; }
static inline unsigned ldg_u8(const unsigned char *p) { return *p; }
static inline unsigned ldg_text4(const unsigned char *p) { return ((const lbz_text4 *)p)->a; }
static inline void stg_u64(unsigned long long *p, unsigned long long v) { *p = v; }
static inline void stg_u32(unsigned *p, unsigned v) { *p = v; }
#endif
