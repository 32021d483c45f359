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

/* The decoder's bit chain (k_decode.hip, dhuff_block): lane j of `nx` holds the bit offset of the code BEHIND the code
 * that starts at offset j, or j | 64 | flags where the walk must stop (no table entry, or the next code starts outside
 * these 64 offsets; the flags say which and where).  Starting at `start`, follow the offsets to a stop; M collects the
 * offsets visited (the stop included), off is the stop's entry.  A hop is one v_readlane whose lane select (its low six bits) is the previous one's result -- ~31 cycles
 * from result to result whatever stands between them (tests/tools/micro/hops.hip), and a scalar instruction that reads
 * such a result waits for it too, so the offsets are marked ten at a time behind the hops; hopping on from a stop goes
 * nowhere. */
__device__ __forceinline__ void huff_walk(unsigned nx, unsigned start, unsigned &off, unsigned long long &M)   /* off: the stop's lane entry as it is (bits 0..5 = its offset) */
{
  unsigned n1, n2, n3, n4, n5, n6, n7, n8, n9;
  asm volatile(
    "s_mov_b64 %[M], 0\n\t"
    "s_mov_b32 %[off], %[start]\n"
    "HW_LOOP_%=:\n\t"
    "v_readlane_b32 %[n1], %[nx], %[off]\n\t"
    "s_bitset1_b64 %[M], %[off]\n\t"
    "s_nop 2\n\t"
    "v_readlane_b32 %[n2], %[nx], %[n1]\n\t"
    "s_nop 3\n\t"
    "v_readlane_b32 %[n3], %[nx], %[n2]\n\t"
    "s_nop 3\n\t"
    "v_readlane_b32 %[n4], %[nx], %[n3]\n\t"
    "s_nop 3\n\t"
    "v_readlane_b32 %[n5], %[nx], %[n4]\n\t"
    "s_nop 3\n\t"
    "v_readlane_b32 %[n6], %[nx], %[n5]\n\t"
    "s_nop 3\n\t"
    "v_readlane_b32 %[n7], %[nx], %[n6]\n\t"
    "s_nop 3\n\t"
    "v_readlane_b32 %[n8], %[nx], %[n7]\n\t"
    "s_nop 3\n\t"
    "v_readlane_b32 %[n9], %[nx], %[n8]\n\t"
    "s_nop 3\n\t"
    "v_readlane_b32 %[off], %[nx], %[n9]\n\t"
    "s_bitset1_b64 %[M], %[n1]\n\t"
    "s_bitset1_b64 %[M], %[n2]\n\t"
    "s_bitset1_b64 %[M], %[n3]\n\t"
    "s_bitset1_b64 %[M], %[n4]\n\t"
    "s_bitset1_b64 %[M], %[n5]\n\t"
    "s_bitset1_b64 %[M], %[n6]\n\t"
    "s_bitset1_b64 %[M], %[n7]\n\t"
    "s_bitset1_b64 %[M], %[n8]\n\t"
    "s_bitset1_b64 %[M], %[n9]\n\t"
    "s_bitcmp0_b32 %[off], 6\n\t"
    "s_cbranch_scc1 HW_LOOP_%=\n\t"
    "s_bitset1_b64 %[M], %[off]"
    : [off] "=&s"(off), [M] "=&s"(M), [n1] "=&s"(n1), [n2] "=&s"(n2), [n3] "=&s"(n3), [n4] "=&s"(n4), [n5] "=&s"(n5), [n6] "=&s"(n6), [n7] "=&s"(n7), [n8] "=&s"(n8), [n9] "=&s"(n9)
    : [nx] "v"(nx), [start] "s"(start)
    : "scc");
}

/* The lanes of M store the symbol of their table entry (e >> 5) side by side from sym16[at] on: the set of lanes IS the
 * execution mask, a lane's place is the number of set bits below it. */
__device__ __forceinline__ void huff_store(unsigned short *sym16, unsigned at, unsigned e, unsigned long long M)
{
  unsigned r, d;
  asm volatile(
    "s_mov_b64 exec, %[M]\n\t"
    "v_mbcnt_lo_u32_b32 %[r], %[Mlo], 0\n\t"
    "v_mbcnt_hi_u32_b32 %[r], %[Mhi], %[r]\n\t"
    "v_lshrrev_b32 %[d], 5, %[e]\n\t"
    "v_add_lshl_u32 %[r], %[r], %[at], 1\n\t"
    "global_store_short %[r], %[d], %[base]\n\t"
    "s_mov_b64 exec, -1"
    : [r] "=&v"(r), [d] "=&v"(d)
    : [M] "s"(M), [Mlo] "s"((unsigned)M), [Mhi] "s"((unsigned)(M >> 32)), [e] "v"(e), [at] "s"(at), [base] "s"(sym16)
    : "memory");
}

/* The decoder's move-to-front chain over one strip of 64 symbols (k_decode.hip, dmtf_chunks): for every lane i of m, in
 * order, list entry v[i] - 1 moves to the front and outv[i] = the entry + 2.  The list is L0..L3 (entry k in lane k & 63
 * of register k >> 6): a front move from entry nn < 64 is one wave_shr of L0 under a lane mask; from further back the
 * registers below the entry's shift whole and their last lanes carry over.  Hand-scheduled for the same reason as
 * huff_walk: one wave, one dependent chain.                                                                   */
__device__ __forceinline__ void mtf_strip(int &L0, int &L1, int &L2, int &L3, unsigned v, unsigned long long m, int &outv, unsigned lane)
{
  unsigned i, nn, x, t0, t2, vsh;
  asm volatile(
    "s_cmp_eq_u64 %[m], 0\n\t"
    "s_cbranch_scc1 MS_END_%=\n"
    "MS_LOOP_%=:\n\t"
    "s_ff1_i32_b64 %[i], %[m]\n\t"
    "s_bitset0_b64 %[m], %[i]\n\t"
    "v_readlane_b32 %[nn], %[v], %[i]\n\t"
    "s_mov_b32 m0, %[i]\n\t"
    "s_sub_u32 %[nn], %[nn], 1\n\t"
    "s_cmp_ge_u32 %[nn], 64\n\t"
    "s_cbranch_scc1 MS_FAR_%=\n\t"
    "v_readlane_b32 %[x], %[L0], %[nn]\n\t"
    "v_mov_b32_dpp %[vsh], %[L0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_cmp_ge_u32 vcc, %[nn], %[lane]\n\t"
    "s_nop 1\n\t"                               /* VCC is an SGPR pair: 2 wait states before a VALU reads what a VALU wrote */
    "v_cndmask_b32 %[L0], %[L0], %[vsh], vcc\n\t"
    "v_writelane_b32 %[L0], %[x], 0\n"
    "MS_OUT_%=:\n\t"
    "s_add_u32 %[x], %[x], 2\n\t"
    "s_cmp_lg_u64 %[m], 0\n\t"
    "v_writelane_b32 %[outv], %[x], m0\n\t"
    "s_cbranch_scc1 MS_LOOP_%=\n\t"
    "s_branch MS_END_%=\n"
    "MS_FAR_%=:\n\t"
    "s_and_b32 %[t0], %[nn], 63\n\t"
    "v_readlane_b32 %[t2], %[L0], 63\n\t"
    "v_cmp_ge_u32 vcc, %[t0], %[lane]\n\t"
    "s_cmp_ge_u32 %[nn], 128\n\t"
    "s_cbranch_scc1 MS_FAR2_%=\n\t"
    "v_readlane_b32 %[x], %[L1], %[t0]\n\t"
    "v_mov_b32_dpp %[vsh], %[L1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_cndmask_b32 %[L1], %[L1], %[vsh], vcc\n\t"
    "v_writelane_b32 %[L1], %[t2], 0\n\t"
    "s_branch MS_FAR0_%=\n"
    "MS_FAR2_%=:\n\t"
    "v_readlane_b32 s42, %[L1], 63\n\t"
    "v_mov_b32_dpp %[L1], %[L1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_writelane_b32 %[L1], %[t2], 0\n\t"
    "s_cmp_ge_u32 %[nn], 192\n\t"
    "s_cbranch_scc1 MS_FAR3_%=\n\t"
    "v_readlane_b32 %[x], %[L2], %[t0]\n\t"
    "v_mov_b32_dpp %[vsh], %[L2] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_cndmask_b32 %[L2], %[L2], %[vsh], vcc\n\t"
    "v_writelane_b32 %[L2], s42, 0\n\t"
    "s_branch MS_FAR0_%=\n"
    "MS_FAR3_%=:\n\t"
    "v_readlane_b32 s43, %[L2], 63\n\t"
    "v_mov_b32_dpp %[L2], %[L2] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_writelane_b32 %[L2], s42, 0\n\t"
    "v_readlane_b32 %[x], %[L3], %[t0]\n\t"
    "v_mov_b32_dpp %[vsh], %[L3] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_cndmask_b32 %[L3], %[L3], %[vsh], vcc\n\t"
    "v_writelane_b32 %[L3], s43, 0\n"
    "MS_FAR0_%=:\n\t"
    "v_mov_b32_dpp %[L0], %[L0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_writelane_b32 %[L0], %[x], 0\n\t"
    "s_branch MS_OUT_%=\n"
    "MS_END_%=:"
    : [L0] "+v"(L0), [L1] "+v"(L1), [L2] "+v"(L2), [L3] "+v"(L3), [outv] "+v"(outv), [m] "+s"(m), [i] "=&s"(i), [nn] "=&s"(nn), [x] "=&s"(x),
      [t0] "=&s"(t0), [t2] "=&s"(t2), [vsh] "=&v"(vsh)
    : [v] "v"(v), [lane] "v"(lane)
    : "s42", "s43", "m0", "vcc", "scc");
}

/* A counter in LDS that one wave of the workgroup advances and the others wait for: publishing makes the wave's earlier
 * stores visible to the workgroup first; what is observed is wave-uniform.  wave_pause() is what a waiting wave does
 * between two looks. */
__device__ __forceinline__ void lds_publish(unsigned *p, unsigned v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned lds_observe(unsigned *p)
{
  return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void wave_pause() { __builtin_amdgcn_s_sleep(4); }

/* wave_shl:1 -- lane l of the result is lane l + 1 of v (lane 63: v's own) */
__device__ __forceinline__ int wave_shl1(int v)
{
  return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);
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
