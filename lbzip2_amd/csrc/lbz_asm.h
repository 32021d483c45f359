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

/* The common cases of the decoder's symbol loop (k_decode.hip, dhuff_block), hand-scheduled: a lone wave
 * issues one dependent instruction every ~2.5 ns whatever its kind, so the loop is as short as it can be made
 * and stays in scalar registers.  All arguments are wave-uniform except L0..L3 (the move-to-front list, entry i in
 * lane i & 63 of register i >> 6), cur (the input window, dword i of the current 256-byte chunk in lane i, MSB first) and lane.
 *
 * A strip at a time: with more than 32 bits in the buffer, lane j looks up the 10 bits that start at bit j (ONE LDS
 * read, entry = symbol << 5 | length); the symbols of the strip are then taken with v_readlane at the running bit
 * offset, so the LDS latency (~37 ns) is paid once per ~10 symbols and not per symbol.  Per symbol: RUNA/RUNB add to the pending zero run; any other symbol first stores a pending run of <= 64 bytes, then
 * moves list entry symbol - 1 to the front and stores it (entries 64..255 live in L1..L3).  The loop RETURNS, with nothing of the current
 * symbol consumed, when the group's 50 symbols are done, the table has no entry (long code, end of block),
 * the run is longer than 64 or would overflow, or the next dword is the
 * last of its chunk -- the caller takes one general step and comes back.                               */
__device__ __forceinline__ void huff_fast(unsigned long long &buf, unsigned &live, unsigned &dwl, unsigned cur, unsigned &k,
                                          unsigned &n, unsigned &es, unsigned &N, int &L0, int &L1, int &L2, int &L3,
                                          unsigned lutaddr, unsigned char *tt8, unsigned maxn, unsigned lane)
{
  unsigned t0, t1, t2, e, l, sym, off, lim, va, vb, vsh, ve;
  const unsigned vneg = (32u - lane) & 31u;
  asm volatile(
    "s_mov_b64 s[40:41], %[buf]\n\t"
    "s_mov_b32 s46, 1\n\t"                      /* lane 0 */
    "s_mov_b32 s47, 0\n\t"
    "s_mov_b32 s48, 0\n\t"                      /* lanes 32..63 */
    "s_mov_b32 s49, -1\n"
    /* a strip: every lane looks up the 10 bits that start at ITS bit offset of the buffer, one LDS read for all */
    "HF_REFRESH_%=:\n\t"
    "s_cmp_gt_u32 %[live], 32\n\t"
    "s_cbranch_scc1 HF_STRIP_%=\n\t"
    "s_and_b32 %[t0], %[dwl], 63\n\t"
    "s_cmp_eq_u32 %[t0], 63\n\t"
    "s_cbranch_scc1 HF_DONE_%=\n\t"
    "v_readlane_b32 s42, %[cur], %[t0]\n\t"
    "s_mov_b32 s43, 0\n\t"
    "s_sub_u32 %[t0], 32, %[live]\n\t"
    "s_lshl_b64 s[42:43], s[42:43], %[t0]\n\t"
    "s_or_b64 s[40:41], s[40:41], s[42:43]\n\t"
    "s_add_u32 %[live], %[live], 32\n\t"
    "s_add_u32 %[dwl], %[dwl], 1\n"
    "HF_STRIP_%=:\n\t"
    "v_mov_b32 %[vb], s40\n\t"
    "v_mov_b32 %[vsh], s41\n\t"
    "v_alignbit_b32 %[va], s41, %[vb], %[vneg]\n\t"          /* lanes 1..31: (hi << j) | (lo >> (32 - j)) */
    "v_cndmask_b32_e64 %[va], %[va], %[vsh], s[46:47]\n\t"   /* lane 0: hi */
    "v_lshlrev_b32 %[vb], %[lane], %[vb]\n\t"                /* lanes 32..63: lo << (j - 32) */
    "v_cndmask_b32_e64 %[va], %[va], %[vb], s[48:49]\n\t"
    "v_lshrrev_b32 %[va], 21, %[va]\n\t"
    "v_and_b32 %[va], 0x7fe, %[va]\n\t"
    "v_add_u32 %[va], %[lut], %[va]\n\t"
    "ds_read_u16 %[ve], %[va]\n\t"
    "s_mov_b32 %[off], 0\n\t"
    "s_sub_u32 %[lim], %[live], 10\n\t"
    "s_min_u32 %[lim], %[lim], 53\n\t"        /* a strip ends before bit 64: the 64-bit shift that drops it takes 0..63 */
    "s_waitcnt lgkmcnt(0)\n"
    "HF_LOOP_%=:\n\t"
    "s_cmp_ge_u32 %[k], 50\n\t"
    "s_cbranch_scc1 HF_EXIT_%=\n\t"
    "s_cmp_gt_u32 %[off], %[lim]\n\t"
    "s_cbranch_scc1 HF_CONSUME_%=\n\t"
    "v_readlane_b32 %[e], %[ve], %[off]\n\t"
    "s_cmp_eq_u32 %[e], 0\n\t"
    "s_cbranch_scc1 HF_EXIT_%=\n\t"
    "s_and_b32 %[l], %[e], 31\n\t"
    "s_lshr_b32 %[sym], %[e], 5\n\t"
    "s_cmp_le_u32 %[sym], 1\n\t"
    "s_cbranch_scc1 HF_RUN_%=\n\t"
    "s_sub_u32 %[sym], %[sym], 1\n\t"
    "s_cmp_eq_u32 %[es], 0\n\t"
    "s_cbranch_scc1 HF_LIT_%=\n\t"
    "s_cmp_gt_u32 %[es], 64\n\t"
    "s_cbranch_scc1 HF_EXIT_%=\n\t"
    "s_add_u32 %[t0], %[n], %[es]\n\t"
    "s_cmp_gt_u32 %[t0], %[maxn]\n\t"
    "s_cbranch_scc1 HF_EXIT_%=\n\t"
    "v_readlane_b32 %[t1], %[L0], 0\n\t"
    "v_add_u32 %[va], %[n], %[lane]\n\t"
    "s_nop 0\n\t"                                /* 2 wait states before a VALU reads the SGPR a VALU wrote */
    "v_mov_b32 %[vb], %[t1]\n\t"
    "v_cmp_gt_u32 vcc, %[es], %[lane]\n\t"
    "s_and_saveexec_b64 s[44:45], vcc\n\t"
    "global_store_byte %[va], %[vb], %[tt8]\n\t"
    "s_mov_b64 exec, s[44:45]\n\t"
    "s_mov_b32 %[n], %[t0]\n\t"
    "s_mov_b32 %[es], 0\n\t"
    "s_mov_b32 %[N], 0\n"
    "HF_LIT_%=:\n\t"
    "s_add_u32 %[off], %[off], %[l]\n\t"
    "s_add_u32 %[k], %[k], 1\n\t"
    "s_cmp_ge_u32 %[sym], 64\n\t"
    "s_cbranch_scc1 HF_FAR_%=\n\t"
    "v_readlane_b32 %[t1], %[L0], %[sym]\n\t"
    "v_mov_b32_dpp %[vsh], %[L0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_cmp_ge_u32 vcc, %[sym], %[lane]\n\t"
    "s_nop 1\n\t"                                /* VCC is an SGPR pair: the same 2 wait states */
    "v_cndmask_b32 %[L0], %[L0], %[vsh], vcc\n\t"
    "v_writelane_b32 %[L0], %[t1], 0\n\t"
    "HF_OUT_%=:\n\t"
    "v_mov_b32 %[vb], %[t1]\n\t"
    "v_mov_b32 %[va], %[n]\n\t"
    "global_store_byte %[va], %[vb], %[tt8]\n\t"
    "s_add_u32 %[n], %[n], 1\n\t"
    "s_branch HF_LOOP_%=\n"
    /* a front move from entry 64..255: registers below the entry's shift whole, their last lanes carry over */
    "HF_FAR_%=:\n\t"
    "s_and_b32 %[t0], %[sym], 63\n\t"
    "v_readlane_b32 %[t2], %[L0], 63\n\t"
    "v_cmp_ge_u32 vcc, %[t0], %[lane]\n\t"
    "s_cmp_ge_u32 %[sym], 128\n\t"
    "s_cbranch_scc1 HF_FAR2_%=\n\t"
    "v_readlane_b32 %[t1], %[L1], %[t0]\n\t"
    "v_mov_b32_dpp %[vsh], %[L1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_cndmask_b32 %[L1], %[L1], %[vsh], vcc\n\t"
    "v_writelane_b32 %[L1], %[t2], 0\n\t"
    "s_branch HF_FAR0_%=\n"
    "HF_FAR2_%=:\n\t"
    "v_readlane_b32 s42, %[L1], 63\n\t"
    "v_mov_b32_dpp %[L1], %[L1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_writelane_b32 %[L1], %[t2], 0\n\t"
    "s_cmp_ge_u32 %[sym], 192\n\t"
    "s_cbranch_scc1 HF_FAR3_%=\n\t"
    "v_readlane_b32 %[t1], %[L2], %[t0]\n\t"
    "v_mov_b32_dpp %[vsh], %[L2] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_cndmask_b32 %[L2], %[L2], %[vsh], vcc\n\t"
    "v_writelane_b32 %[L2], s42, 0\n\t"
    "s_branch HF_FAR0_%=\n"
    "HF_FAR3_%=:\n\t"
    "v_readlane_b32 s43, %[L2], 63\n\t"
    "v_mov_b32_dpp %[L2], %[L2] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_writelane_b32 %[L2], s42, 0\n\t"
    "v_readlane_b32 %[t1], %[L3], %[t0]\n\t"
    "v_mov_b32_dpp %[vsh], %[L3] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_cndmask_b32 %[L3], %[L3], %[vsh], vcc\n\t"
    "v_writelane_b32 %[L3], s43, 0\n"
    "HF_FAR0_%=:\n\t"
    "v_mov_b32_dpp %[L0], %[L0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "v_writelane_b32 %[L0], %[t1], 0\n\t"
    "s_branch HF_OUT_%=\n"
    "HF_RUN_%=:\n\t"
    "s_cmp_ge_u32 %[N], 21\n\t"
    "s_cbranch_scc1 HF_EXIT_%=\n\t"
    "s_add_u32 %[t0], %[sym], 1\n\t"
    "s_lshl_b32 %[t0], %[t0], %[N]\n\t"
    "s_add_u32 %[es], %[es], %[t0]\n\t"
    "s_add_u32 %[N], %[N], 1\n\t"
    "s_add_u32 %[off], %[off], %[l]\n\t"
    "s_add_u32 %[k], %[k], 1\n\t"
    "s_branch HF_LOOP_%=\n"
    /* the strip is used up: drop its bits, look the next ones up */
    "HF_CONSUME_%=:\n\t"
    "s_lshl_b64 s[40:41], s[40:41], %[off]\n\t"
    "s_sub_u32 %[live], %[live], %[off]\n\t"
    "s_branch HF_REFRESH_%=\n"
    "HF_EXIT_%=:\n\t"
    "s_lshl_b64 s[40:41], s[40:41], %[off]\n\t"
    "s_sub_u32 %[live], %[live], %[off]\n"
    "HF_DONE_%=:\n\t"
    "s_mov_b64 %[buf], s[40:41]"
    : [buf] "+s"(buf), [live] "+s"(live), [dwl] "+s"(dwl), [k] "+s"(k), [n] "+s"(n), [es] "+s"(es), [N] "+s"(N), [L0] "+v"(L0),
      [L1] "+v"(L1), [L2] "+v"(L2), [L3] "+v"(L3), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [e] "=&s"(e), [l] "=&s"(l), [sym] "=&s"(sym),
      [off] "=&s"(off), [lim] "=&s"(lim), [va] "=&v"(va), [vb] "=&v"(vb), [vsh] "=&v"(vsh), [ve] "=&v"(ve)
    : [cur] "v"(cur), [lane] "v"(lane), [vneg] "v"(vneg), [lut] "s"(lutaddr), [tt8] "s"(tt8), [maxn] "s"(maxn)
    : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "vcc", "scc", "memory");
}

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
