// Microbenchmark (developer tool), round 4: what ONE wave pays per instruction when the next instruction depends
// on it - the regime the compositing kernels run in (profiles/r04c_ablate_occupancy.txt: a wave alone on its SIMD
// takes 174 us for raster_bwd's work, with three neighbours 206 us, so it is the wave's own dependency chain, not
// the shared pipe, that sets the time).  One wave per SIMD (160 KiB of LDS per 256-thread workgroup), s_memtime
// around 4000 instructions; "ILP k" = k independent chains interleaved.
// build: hipcc --offload-arch=gfx950 -O3 -o lat_bench tools/micro/lat_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

enum { L_FMA1 = 0, L_FMA2, L_FMA4, L_FMA8, L_FMAC1, L_FMAC2, L_MUL_ADD1, L_EXP1, L_EXP2, L_EXP4, L_RCP1, L_DPP1, L_DPP2,
       L_CMP_CND, L_CMP_SAND_CND, L_CMP_BRANCH, L_BRANCH_TAKEN, L_BRANCH_NOT, L_READFIRST_SALU, L_BPERM1, L_BPERM4,
       L_DSREAD128, L_DSREAD128_X3, L_SWAP16, L_SWAP32, L_MOVDPP_SNOP, L_FMA_SGPRDEP, L_NOP, L_CND_VCC_ONLY, L_SAND_CND_VCC, L_SAND_CND_SGPR, L_CMP_CND_VCC, L_CMP2_SAND_CND, L_CMP2_CND2, L_COUNT };

template <int MODE>
__global__ void k(float* out, unsigned long long* times, int iters, float s0, float s1) {
    extern __shared__ float lds[];
    const float t = (float)threadIdx.x * 1e-3f;
    float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
    float m = s0 + t * 1e-9f, c = s1;
    asm volatile("" : "+v"(m), "+v"(c));
    int addr = ((threadIdx.x & 63) ^ 16) * 4;
    int laddr = (threadIdx.x >> 6) * 1024;          // wave-uniform LDS address (broadcast read)
    lds[threadIdx.x] = t;
    __syncthreads();
    unsigned long long sm = 0x5555555555555555ull;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (MODE == L_FMA1) {
                asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                             "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                             : "+v"(a0) : "v"(m), "v"(c));
            } else if (MODE == L_FMA2) {
                asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\t"
                             "v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3"
                             : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));
            } else if (MODE == L_FMA4) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
                             "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
            } else if (MODE == L_FMA8) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                             "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (MODE == L_FMAC1) {
                asm volatile("v_fmac_f32_e32 %0, %0, %1\n\tv_fmac_f32_e32 %0, %0, %1\n\tv_fmac_f32_e32 %0, %0, %1\n\tv_fmac_f32_e32 %0, %0, %1\n\t"
                             "v_fmac_f32_e32 %0, %0, %1\n\tv_fmac_f32_e32 %0, %0, %1\n\tv_fmac_f32_e32 %0, %0, %1\n\tv_fmac_f32_e32 %0, %0, %1"
                             : "+v"(a0) : "v"(m));
            } else if (MODE == L_FMAC2) {
                asm volatile("v_fmac_f32_e32 %0, %0, %2\n\tv_fmac_f32_e32 %1, %1, %2\n\tv_fmac_f32_e32 %0, %0, %2\n\tv_fmac_f32_e32 %1, %1, %2\n\t"
                             "v_fmac_f32_e32 %0, %0, %2\n\tv_fmac_f32_e32 %1, %1, %2\n\tv_fmac_f32_e32 %0, %0, %2\n\tv_fmac_f32_e32 %1, %1, %2"
                             : "+v"(a0), "+v"(a1) : "v"(m));
            } else if (MODE == L_MUL_ADD1) {
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\t"
                             "v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2"
                             : "+v"(a0) : "v"(m), "v"(c));
            } else if (MODE == L_EXP1) {
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\t"
                             "v_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0" : "+v"(a0));
            } else if (MODE == L_EXP2) {
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\t"
                             "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(a0), "+v"(a1));
            } else if (MODE == L_EXP4) {
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                             "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else if (MODE == L_RCP1) {
                asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\t"
                             "v_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0" : "+v"(a0));
            } else if (MODE == L_DPP1) {     // dependent DPP adds need 2 wait states: the assembler does not add them in asm
                asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0));
            } else if (MODE == L_DPP2) {
                asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 0\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 0\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 0\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1));
            } else if (MODE == L_CMP_CND) {      // compare -> select through an SGPR pair, chain of 4 pairs
                asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]\n\t"
                             "v_cmp_le_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]\n\t"
                             "v_cmp_le_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]\n\t"
                             "v_cmp_le_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]"
                             : "+v"(a0) : "v"(m) : "s20", "s21");
            } else if (MODE == L_CMP_SAND_CND) { // compare -> s_and -> select (what a two-condition body does), 4 triples
                asm volatile("v_cmp_le_f32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t"
                             "v_cmp_le_f32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t"
                             "v_cmp_le_f32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t"
                             "v_cmp_le_f32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc"
                             : "+v"(a0) : "v"(m), "s"(sm) : "vcc", "scc");
            } else if (MODE == L_CMP_BRANCH) {   // compare -> scalar test -> branch not taken, 4 times
                asm volatile("v_cmp_le_f32 vcc, %0, %1\n\ts_cmp_eq_u64 vcc, 0\n\ts_cbranch_scc1 1f\n\tv_add_f32 %0, %0, %1\n1:\n\t"
                             "v_cmp_le_f32 vcc, %0, %1\n\ts_cmp_eq_u64 vcc, 0\n\ts_cbranch_scc1 2f\n\tv_add_f32 %0, %0, %1\n2:\n\t"
                             "v_cmp_le_f32 vcc, %0, %1\n\ts_cmp_eq_u64 vcc, 0\n\ts_cbranch_scc1 3f\n\tv_add_f32 %0, %0, %1\n3:\n\t"
                             "v_cmp_le_f32 vcc, %0, %1\n\ts_cmp_eq_u64 vcc, 0\n\ts_cbranch_scc1 4f\n\tv_add_f32 %0, %0, %1\n4:"
                             : "+v"(a0) : "v"(m) : "vcc", "scc");
            } else if (MODE == L_BRANCH_TAKEN) {  // 8 taken forward branches over one instruction each
                asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\tv_add_f32 %0, %0, %1\n1:\n\ts_cbranch_scc1 2f\n\tv_add_f32 %0, %0, %1\n2:\n\t"
                             "s_cbranch_scc1 3f\n\tv_add_f32 %0, %0, %1\n3:\n\ts_cbranch_scc1 4f\n\tv_add_f32 %0, %0, %1\n4:\n\t"
                             "s_cbranch_scc1 5f\n\tv_add_f32 %0, %0, %1\n5:\n\ts_cbranch_scc1 6f\n\tv_add_f32 %0, %0, %1\n6:\n\t"
                             "s_cbranch_scc1 7f\n\tv_add_f32 %0, %0, %1\n7:\n\ts_cbranch_scc1 8f\n\tv_add_f32 %0, %0, %1\n8:"
                             : "+v"(a0) : "v"(m) : "scc");
            } else if (MODE == L_BRANCH_NOT) {    // 8 untaken branches
                asm volatile("s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n1:\n\ts_cbranch_scc1 2f\n2:\n\ts_cbranch_scc1 3f\n3:\n\ts_cbranch_scc1 4f\n4:\n\t"
                             "s_cbranch_scc1 5f\n5:\n\ts_cbranch_scc1 6f\n6:\n\ts_cbranch_scc1 7f\n7:\n\ts_cbranch_scc1 8f\n8:" : : : "scc");
            } else if (MODE == L_READFIRST_SALU) { // VGPR -> SGPR -> scalar op -> VGPR, 4 round trips
                asm volatile("v_readfirstlane_b32 s20, %0\n\ts_add_i32 s20, s20, 1\n\tv_mov_b32 %0, s20\n\t"
                             "v_readfirstlane_b32 s20, %0\n\ts_add_i32 s20, s20, 1\n\tv_mov_b32 %0, s20\n\t"
                             "v_readfirstlane_b32 s20, %0\n\ts_add_i32 s20, s20, 1\n\tv_mov_b32 %0, s20\n\t"
                             "v_readfirstlane_b32 s20, %0\n\ts_add_i32 s20, s20, 1\n\tv_mov_b32 %0, s20"
                             : "+v"(a0) : : "s20", "scc");
            } else if (MODE == L_BPERM1) {        // 2 dependent ds_bpermute round trips
                asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\tds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(a0) : "v"(addr));
            } else if (MODE == L_BPERM4) {        // 4 independent bpermutes, one wait
                asm volatile("ds_bpermute_b32 %0, %4, %0\n\tds_bpermute_b32 %1, %4, %1\n\tds_bpermute_b32 %2, %4, %2\n\t"
                             "ds_bpermute_b32 %3, %4, %3\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(addr));
            } else if (MODE == L_DSREAD128) {     // 2 dependent uniform-address ds_read_b128 (address from the data)
                asm volatile("ds_read_b128 v[40:43], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xff0, v40\n\t"
                             "ds_read_b128 v[40:43], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xff0, v40"
                             : "+v"(laddr) : : "v40", "v41", "v42", "v43");
            } else if (MODE == L_DSREAD128_X3) {  // 3 reads issued together, one wait, then the dependent address
                asm volatile("ds_read_b128 v[40:43], %0\n\tds_read_b128 v[44:47], %0 offset:16\n\tds_read_b128 v[48:51], %0 offset:32\n\t"
                             "s_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xff0, v40"
                             : "+v"(laddr) : : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
            } else if (MODE == L_SWAP16) {
                asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1"
                             : "+v"(a0), "+v"(a1));
            } else if (MODE == L_SWAP32) {
                asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %0, %1"
                             : "+v"(a0), "+v"(a1));
            } else if (MODE == L_FMA_SGPRDEP) {   // VALU writes an SGPR that the next VALU reads as an operand
                asm volatile("v_readfirstlane_b32 s20, %0\n\tv_fma_f32 %0, s20, %1, %0\n\tv_readfirstlane_b32 s20, %0\n\tv_fma_f32 %0, s20, %1, %0\n\t"
                             "v_readfirstlane_b32 s20, %0\n\tv_fma_f32 %0, s20, %1, %0\n\tv_readfirstlane_b32 s20, %0\n\tv_fma_f32 %0, s20, %1, %0"
                             : "+v"(a0) : "v"(m) : "s20");
            } else if (MODE == L_CND_VCC_ONLY) {   // 8 selects on a vcc written once before the loop
                asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t"
                             "v_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t"
                             "v_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a0) : "v"(m) : );
            } else if (MODE == L_SAND_CND_VCC) {   // SALU writes vcc, the next VALU selects on it: 4 pairs
                asm volatile("s_and_b64 vcc, %2, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\ts_and_b64 vcc, %2, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t"
                             "s_and_b64 vcc, %2, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\ts_and_b64 vcc, %2, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc"
                             : "+v"(a0) : "v"(m), "s"(sm) : "vcc", "scc");
            } else if (MODE == L_SAND_CND_SGPR) {  // the same through an ordinary SGPR pair
                asm volatile("s_and_b64 s[20:21], %2, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\ts_and_b64 s[20:21], %2, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\t"
                             "s_and_b64 s[20:21], %2, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\ts_and_b64 s[20:21], %2, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]"
                             : "+v"(a0) : "v"(m), "s"(sm) : "s20", "s21", "scc");
            } else if (MODE == L_CMP_CND_VCC) {    // VALU writes vcc, the next VALU selects on it: 4 pairs
                asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t"
                             "v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc"
                             : "+v"(a0) : "v"(m) : "vcc");
            } else if (MODE == L_CMP2_SAND_CND) {  // the body's pattern: two compares, s_and, one select; 2 groups
                asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n\tv_cmp_le_f32 vcc, %1, %0\n\ts_and_b64 vcc, s[20:21], vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t"
                             "v_cmp_le_f32 s[20:21], %0, %1\n\tv_cmp_le_f32 vcc, %1, %0\n\ts_and_b64 vcc, s[20:21], vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc"
                             : "+v"(a0) : "v"(m) : "vcc", "s20", "s21", "scc");
            } else if (MODE == L_CMP2_CND2) {      // the same decision without the SALU: two compares, two selects; 2 groups
                asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n\tv_cmp_le_f32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\t"
                             "v_cmp_le_f32 s[20:21], %0, %1\n\tv_cmp_le_f32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]"
                             : "+v"(a0) : "v"(m) : "vcc", "s20", "s21");
            } else if (MODE == L_NOP) {
                asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)laddr + lds[(threadIdx.x * 7) & 255];
    if ((threadIdx.x & 63) == 0) times[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char* name, double units_per_rep, const char* unit, float* out, unsigned long long* td, FILE* f) {
    const int iters = 100;
    char line[512];
    int n = snprintf(line, sizeof line, "%-64s", name);
    for (int occ : {1, 2, 4}) {
        const int threads = 256 * occ;
        const size_t lds = 160 * 1024;
        (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int grid = 256, waves = grid * threads / 64;
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), lds, 0, out, td, iters, 1.0001f, 1e-7f);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h(waves);
            (void)hipMemcpy(h.data(), td, waves * 8, hipMemcpyDeviceToHost);
            std::nth_element(h.begin(), h.begin() + waves / 2, h.end());
            best = std::min(best, (double)h[waves / 2]);
        }
        n += snprintf(line + n, sizeof line - n, " | %6.1f", best / (iters * 8.0 * units_per_rep));
    }
    n += snprintf(line + n, sizeof line - n, "  cycles per %s", unit);
    printf("%s\n", line);
    fflush(stdout);
    if (f) { fprintf(f, "%s\n", line); fflush(f); }
}

int main(int argc, char** argv) {
    float* out; (void)hipMalloc(&out, 1024 * 1024 * 4);
    unsigned long long* td; (void)hipMalloc(&td, 8192 * 8);
    FILE* f = argc > 1 ? fopen(argv[1], "w") : nullptr;
    const char* hdr = "cycles (s_memtime) a wave spends per unit, with 1 | 2 | 4 waves resident per SIMD";
    printf("%s\n", hdr); if (f) fprintf(f, "%s\n", hdr);
#define R(M, NAME, U, UNIT) run<M>(NAME, U, UNIT, out, td, f);
    R(L_NOP, "s_nop 0", 8, "instruction")
    R(L_FMA1, "v_fma_f32, dependent chain (ILP 1)", 8, "instruction")
    R(L_FMA2, "v_fma_f32, ILP 2", 8, "instruction")
    R(L_FMA4, "v_fma_f32, ILP 4", 8, "instruction")
    R(L_FMA8, "v_fma_f32, ILP 8", 8, "instruction")
    R(L_FMAC1, "v_fmac_f32_e32 (VOP2), dependent chain", 8, "instruction")
    R(L_FMAC2, "v_fmac_f32_e32 (VOP2), ILP 2", 8, "instruction")
    R(L_MUL_ADD1, "v_mul_f32 -> v_add_f32 dependent chain", 8, "instruction")
    R(L_EXP1, "v_exp_f32, dependent chain", 8, "instruction")
    R(L_EXP2, "v_exp_f32, ILP 2", 8, "instruction")
    R(L_EXP4, "v_exp_f32, ILP 4", 8, "instruction")
    R(L_RCP1, "v_rcp_f32, dependent chain", 8, "instruction")
    R(L_DPP1, "s_nop 1 + v_add_f32_dpp, dependent chain", 8, "DPP add")
    R(L_DPP2, "v_add_f32_dpp, ILP 2 (s_nop 0 between pairs)", 8, "DPP add")
    R(L_CMP_CND, "v_cmp (SGPR pair) -> v_cndmask, dependent", 4, "pair")
    R(L_CND_VCC_ONLY, "v_cndmask_b32_e32 on a vcc set before the loop, dependent", 8, "instruction")
    R(L_SAND_CND_VCC, "s_and_b64 vcc -> v_cndmask_e32 vcc", 4, "pair")
    R(L_SAND_CND_SGPR, "s_and_b64 s[20:21] -> v_cndmask_e64 s[20:21]", 4, "pair")
    R(L_CMP_CND_VCC, "v_cmp vcc -> v_cndmask_e32 vcc, dependent", 4, "pair")
    R(L_CMP2_SAND_CND, "2 x v_cmp -> s_and_b64 vcc -> v_cndmask vcc (the bwd body's decision)", 2, "group")
    R(L_CMP2_CND2, "2 x v_cmp -> 2 x v_cndmask (no SALU)", 2, "group")
    R(L_CMP_SAND_CND, "v_cmp vcc -> s_and_b64 -> v_cndmask vcc, dependent", 4, "triple")
    R(L_CMP_BRANCH, "v_cmp vcc -> s_cmp_eq_u64 -> s_cbranch (not taken) -> v_add", 4, "group")
    R(L_BRANCH_TAKEN, "s_cbranch_scc1 taken over one instruction", 8, "branch")
    R(L_BRANCH_NOT, "s_cbranch_scc1 not taken", 8, "branch")
    R(L_READFIRST_SALU, "v_readfirstlane -> s_add -> v_mov", 4, "round trip")
    R(L_FMA_SGPRDEP, "v_readfirstlane -> v_fma reading that SGPR", 4, "pair")
    R(L_BPERM1, "ds_bpermute_b32 + wait, dependent", 2, "round trip")
    R(L_BPERM4, "4 x ds_bpermute_b32 + one wait", 1, "group of 4")
    R(L_DSREAD128, "ds_read_b128 (uniform address) + wait, dependent", 2, "round trip")
    R(L_DSREAD128_X3, "3 x ds_read_b128 + one wait", 1, "group of 3")
    R(L_SWAP16, "v_permlane16_swap_b32, dependent", 4, "instruction")
    R(L_SWAP32, "v_permlane32_swap_b32, dependent", 4, "instruction")
    if (f) fclose(f);
    return 0;
}
