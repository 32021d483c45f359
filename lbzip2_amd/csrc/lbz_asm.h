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
 * these 64 offsets; the flags say which and where); nx2 = nx o nx (a stop maps to itself).  Starting at `start`, follow the
 * offsets to a stop; M collects the offsets visited (the stop included), off is the stop's entry.  A hop is a v_readlane
 * whose lane select (its low six bits) is an earlier one's result -- ~31 cycles from result to result whatever stands
 * between them (tests/tools/micro/hops.hip), so TWO walks are in flight, over the codes of even and of odd number, each
 * hopping two codes at a time through nx2; a scalar instruction that reads such a result waits for it too, so the offsets
 * are marked a dozen at a time behind the hops.  Hopping on from a stop goes nowhere, and both walks end on the same one. */
__device__ __forceinline__ void huff_walk(unsigned nx, unsigned nx2, unsigned start, unsigned &off, unsigned long long &M)
{
  unsigned a0, a1, a2, a3, a4, a5, b0, b1, b2, b3, b4, b5, b6;
  asm volatile(
    "s_mov_b64 %[M], 0\n\t"
    "s_mov_b32 %[a0], %[start]\n\t"
    "v_readlane_b32 %[b0], %[nx], %[a0]\n\t"
    "s_nop 3\n"
    "HW_LOOP_%=:\n\t"
    "v_readlane_b32 %[a1], %[nx2], %[a0]\n\t"
    "v_readlane_b32 %[b1], %[nx2], %[b0]\n\t"
    "s_nop 2\n\t"
    "v_readlane_b32 %[a2], %[nx2], %[a1]\n\t"
    "v_readlane_b32 %[b2], %[nx2], %[b1]\n\t"
    "s_nop 2\n\t"
    "v_readlane_b32 %[a3], %[nx2], %[a2]\n\t"
    "v_readlane_b32 %[b3], %[nx2], %[b2]\n\t"
    "s_nop 2\n\t"
    "v_readlane_b32 %[a4], %[nx2], %[a3]\n\t"
    "v_readlane_b32 %[b4], %[nx2], %[b3]\n\t"
    "s_nop 2\n\t"
    "v_readlane_b32 %[a5], %[nx2], %[a4]\n\t"
    "v_readlane_b32 %[b5], %[nx2], %[b4]\n\t"
    "s_nop 2\n\t"
    "v_readlane_b32 %[a6], %[nx2], %[a5]\n\t"
    "v_readlane_b32 %[b6], %[nx2], %[b5]\n\t"
    "s_bitset1_b64 %[M], %[a0]\n\t"
    "s_bitset1_b64 %[M], %[b0]\n\t"
    "s_bitset1_b64 %[M], %[a1]\n\t"
    "s_bitset1_b64 %[M], %[b1]\n\t"
    "s_bitset1_b64 %[M], %[a2]\n\t"
    "s_bitset1_b64 %[M], %[b2]\n\t"
    "s_bitset1_b64 %[M], %[a3]\n\t"
    "s_bitset1_b64 %[M], %[b3]\n\t"
    "s_bitset1_b64 %[M], %[a4]\n\t"
    "s_bitset1_b64 %[M], %[b4]\n\t"
    "s_bitset1_b64 %[M], %[a5]\n\t"
    "s_bitset1_b64 %[M], %[b5]\n\t"
    "s_bitcmp1_b32 %[a6], 6\n\t"
    "s_cbranch_scc1 HW_DONE_%=\n\t"
    "s_mov_b32 %[a0], %[a6]\n\t"
    "s_mov_b32 %[b0], %[b6]\n\t"
    "s_branch HW_LOOP_%=\n"
    "HW_DONE_%=:\n\t"
    "s_bitset1_b64 %[M], %[a6]\n\t"
    "s_bitset1_b64 %[M], %[b6]"
    : [a6] "=&s"(off), [M] "=&s"(M), [a0] "=&s"(a0), [a1] "=&s"(a1), [a2] "=&s"(a2), [a3] "=&s"(a3), [a4] "=&s"(a4), [a5] "=&s"(a5), [b0] "=&s"(b0), [b1] "=&s"(b1), [b2] "=&s"(b2), [b3] "=&s"(b3), [b4] "=&s"(b4), [b5] "=&s"(b5), [b6] "=&s"(b6)
    : [nx] "v"(nx), [nx2] "v"(nx2), [start] "s"(start)
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


/* Reserve `amount` (wave-uniform) consecutive places behind a counter (LDS or HBM): returns the first, wave-uniform.
 * Written without a branch for the reason given at wave_claim: every lane takes part in the atomic, lane 0 alone adds. */
__device__ __forceinline__ unsigned wave_reserve(unsigned *counter, unsigned amount)
{
  const unsigned r = atomicAdd(counter, (threadIdx.x & 63u) == 0u ? amount : 0u);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r);
}

/* 16 bytes of the block text at any byte address (global_load_dwordx4: gfx950 takes unaligned addresses) */
struct __attribute__((packed, aligned(1))) lbz_text16 { unsigned long long a, b; };
struct __attribute__((packed, aligned(1))) lbz_text4 { unsigned a; };

/* acc + (a0 < b) + (a1 < b), unsigned 64-bit: two v_cmp_lt_u64 + v_addc_co_u32 pairs in one block (the compiler's form
 * is compare, select, add, and it separates two blocks that both write VCC with a wait state) */
__device__ __forceinline__ unsigned add_if_less2(unsigned acc, unsigned long long a0, unsigned long long a1, unsigned long long b)
{
  asm("v_cmp_lt_u64 vcc, %1, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
      "v_cmp_lt_u64 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(acc) : "v"(a0), "v"(a1), "v"(b) : "vcc");
  return acc;
}

/* Loads and stores that say "HBM" (address space 1): a pointer that reaches a function through a select or a table of pointers
 * is generic to the compiler, which then issues flat_load -- and a flat load in flight makes every wait for an LDS read a wait
 * for the load as well (both count in lgkmcnt), which undoes any prefetch across LDS work. */
#define LBZ_GLOBAL(T) __attribute__((address_space(1))) T
__device__ __forceinline__ unsigned long long ldg_u64(const unsigned long long *p) { return *(const LBZ_GLOBAL(unsigned long long) *)p; }
__device__ __forceinline__ unsigned ldg_u32(const unsigned *p) { return *(const LBZ_GLOBAL(unsigned) *)p; }
__device__ __forceinline__ unsigned ldg_u8(const unsigned char *p) { return *(const LBZ_GLOBAL(unsigned char) *)p; }
__device__ __forceinline__ unsigned ldg_text4(const unsigned char *p) { return ((const LBZ_GLOBAL(lbz_text4) *)p)->a; }      /* four bytes at any address */
__device__ __forceinline__ void stg_u64(unsigned long long *p, unsigned long long v) { *(LBZ_GLOBAL(unsigned long long) *)p = v; }
__device__ __forceinline__ void stg_u32(unsigned *p, unsigned v) { *(LBZ_GLOBAL(unsigned) *)p = v; }

#endif
