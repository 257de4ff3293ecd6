// Microbenchmark (developer tool), round 4: issue cost of VALU instruction forms in SHADER cycles.
//
// tools/micro/op_bench.hip divides wall time by a nominal 2.4 GHz; its v_fma_f32 figure (3.19) disagrees
// with the guide's 2 cycles (MI355X_MICROARCH.md "Per-instruction cycle constants").  This version separates
// the two unknowns: every wave brackets its loop with s_memtime (shader clock) and s_memrealtime (100 MHz
// constant clock), so that
//   cycles / instruction / SIMD = d(memtime) * waves_per_simd_resident / (instructions per wave)
//   shader clock under this load = d(memtime) / d(memrealtime) * 100 MHz
// are reported independently, for 1, 2, 4 and 8 resident waves per SIMD (occupancy pinned with dynamic LDS:
// one 160 KiB workgroup per CU of 256/512/1024 threads, or two 80 KiB ones of 1024).
// 64 independent instances per loop iteration (4 x 16 registers), so loop overhead is < 5 % of the issue slots.
// The last section interleaves v_mfma_f32_16x16x4_f32 with a VALU stream: does the matrix pipe run beside it?
// build: hipcc --offload-arch=gfx950 -O3 -o op_bench2 tools/micro/op_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

typedef float f4 __attribute__((ext_vector_type(4)));

enum {
    M_FMA = 0, M_FMA_DST, M_FMAC, M_FMA_SGPR, M_MUL, M_ADD, M_ADD_LIT, M_CND_VCC, M_CND_SGPR, M_CMP_VCC, M_CMP_SGPR,
    M_EXP, M_EXP_NEG, M_RCP, M_MAX, M_MOV, M_DPP_QUAD, M_DPP_ROR, M_SUB_ABS, M_FMA_NEG, M_PK_FMA, M_PK_MUL, M_ADD_U32,
    M_AND, M_FMA_EXP_4_1, M_FMA_SALU, M_FMA_MFMA_16_1, M_FMA_MFMA_8_1, M_MFMA_ONLY, M_FMA_MFMA_4_1, M_COUNT
};

template <int MODE>
__global__ void k(float* out, unsigned long long* times, int iters, float s0, float s1) {
    extern __shared__ float lds_pad[];
    float a[16];
    const float t = (float)threadIdx.x * 1e-3f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = t + i;
    float m = s0 + t * 1e-9f, c = s1, d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    asm volatile("" : "+v"(m), "+v"(c));
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    unsigned long long sm = threadIdx.x & 1 ? 0x5555555555555555ull : 0xAAAAAAAAAAAAAAAAull;
    sm = __builtin_amdgcn_readfirstlane((unsigned)sm) |
         ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(sm >> 32)) << 32);
    int sacc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MODE == M_CND_VCC) asm volatile("s_mov_b64 vcc, %0" : : "s"(sm) : "vcc");
#define OP(i)                                                                                                        \
            if (MODE == M_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                    \
            else if (MODE == M_FMA_DST) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d[i]) : "v"(a[i]), "v"(m), "v"(c)); \
            else if (MODE == M_FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));            \
            else if (MODE == M_FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(c));         \
            else if (MODE == M_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));                           \
            else if (MODE == M_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                           \
            else if (MODE == M_ADD_LIT) asm volatile("v_add_f32 %0, 0x3f800347, %0" : "+v"(a[i]));                        \
            else if (MODE == M_CND_VCC) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));          \
            else if (MODE == M_CND_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(sm));     \
            else if (MODE == M_CMP_VCC) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");           \
            else if (MODE == M_CMP_SGPR) asm volatile("v_cmp_le_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(m) : "s20", "s21"); \
            else if (MODE == M_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                        \
            else if (MODE == M_EXP_NEG) asm volatile("v_exp_f32_e64 %0, -%0" : "+v"(a[i]));                               \
            else if (MODE == M_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));                                        \
            else if (MODE == M_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                           \
            else if (MODE == M_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));                               \
            else if (MODE == M_DPP_QUAD) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i])); \
            else if (MODE == M_DPP_ROR) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i])); \
            else if (MODE == M_SUB_ABS) asm volatile("v_sub_f32 %0, |%0|, |%1|" : "+v"(a[i]) : "v"(c));                   \
            else if (MODE == M_FMA_NEG) asm volatile("v_fma_f32 %0, -%0, %1, %1" : "+v"(a[i]) : "v"(m));                  \
            else if (MODE == M_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&d[0]), "v"(*(double*)&d[2])); \
            else if (MODE == M_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&d[0])); \
            else if (MODE == M_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));                       \
            else if (MODE == M_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));                           \
            else if (MODE == M_FMA_EXP_4_1) {                                                                            \
                if ((i & 3) == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                          \
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                              \
            } else if (MODE == M_FMA_SALU) {                                                                             \
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                                   \
                if ((i & 3) == 3) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sacc));                                       \
            } else if (MODE == M_FMA_MFMA_16_1) {                                                                        \
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                                   \
                if (i == 15) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(rep & 1 ? acc1 : acc0) : "v"(m), "v"(c)); \
            } else if (MODE == M_FMA_MFMA_8_1) {                                                                         \
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                                   \
                if ((i & 7) == 7) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(i & 8 ? acc1 : acc0) : "v"(m), "v"(c)); \
            } else if (MODE == M_FMA_MFMA_4_1) {                                                                         \
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                                   \
                if ((i & 3) == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(i & 4 ? acc1 : acc0) : "v"(m), "v"(c)); \
            } else if (MODE == M_MFMA_ONLY) {                                                                            \
                if ((i & 3) == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(i & 4 ? acc1 : acc0) : "v"(m), "v"(c)); \
            }
            REP16(OP)
#undef OP
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = (float)sacc + acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i] + d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + lds_pad[threadIdx.x & 7];
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        times[2 * w] = t1 - t0;
        times[2 * w + 1] = r1 - r0;
    }
}

struct Mode { int id; const char* name; double valu_per_iter; double mfma_per_iter; };

template <int MODE>
static void run(const Mode& md, float* out, unsigned long long* times_d, FILE* f) {
    const int iters = 400;
    char line[512];
    int n = snprintf(line, sizeof line, "%-44s", md.name);
    for (int occ : {1, 2, 4, 8}) {
        const int threads = occ == 1 ? 256 : occ == 2 ? 512 : 1024;
        const int blocks_per_cu = occ == 8 ? 2 : 1;
        const size_t lds = occ == 8 ? 80 * 1024 : 160 * 1024;
        (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int grid = 256 * blocks_per_cu;
        const int waves = grid * threads / 64;
        double best_cyc = 1e30, clk = 0.0, wall_ms = 0.0;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), lds, 0, out, times_d, iters, 1.0001f, 1e-7f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * waves);
            (void)hipMemcpy(h.data(), times_d, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> cyc(waves), rt(waves);
            for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w]; rt[w] = (double)h[2 * w + 1]; }
            std::nth_element(cyc.begin(), cyc.begin() + waves / 2, cyc.end());
            std::nth_element(rt.begin(), rt.begin() + waves / 2, rt.end());
            const double c = cyc[waves / 2];
            if (c < best_cyc) { best_cyc = c; clk = c / rt[waves / 2] * 100e6; wall_ms = ms; }
        }
        const double per_iter = best_cyc * 1.0 / iters;              // shader cycles per loop iteration of ONE wave
        const double insts = md.valu_per_iter > 0 ? md.valu_per_iter : md.mfma_per_iter;
        // occ waves share a SIMD: cycles the SIMD spends per instruction
        n += snprintf(line + n, sizeof line - n, " | %5.2f (%4.2f GHz, %5.3f ms)", per_iter / insts / occ * 1.0 * 1.0, clk * 1e-9, wall_ms);
    }
    printf("%s\n", line);
    if (f) fprintf(f, "%s\n", line);
}

int main(int argc, char** argv) {
    float* out; (void)hipMalloc(&out, 512 * 1024 * 4);
    unsigned long long* times_d; (void)hipMalloc(&times_d, 2 * 8192 * 8);
    FILE* f = argc > 1 ? fopen(argv[1], "w") : nullptr;
    const char* hdr = "SIMD cycles per instruction (s_memtime), by resident waves per SIMD: 1 | 2 | 4 | 8   (shader clock, launch wall time)";
    printf("%s\n", hdr); if (f) fprintf(f, "%s\n", hdr);
#define R(M, NAME, V, MF) { Mode md = {M, NAME, V, MF}; run<M>(md, out, times_d, f); }
    R(M_FMA, "v_fma_f32 d=d*a+b (3 VGPR)", 64, 0)
    R(M_FMA_DST, "v_fma_f32 d'=x*a+b (no RAW chain)", 64, 0)
    R(M_FMAC, "v_fmac_f32_e32 (VOP2)", 64, 0)
    R(M_FMA_SGPR, "v_fma_f32, one SGPR operand", 64, 0)
    R(M_MUL, "v_mul_f32", 64, 0)
    R(M_ADD, "v_add_f32", 64, 0)
    R(M_ADD_LIT, "v_add_f32 literal", 64, 0)
    R(M_CND_VCC, "v_cndmask_b32_e32 vcc (s_mov vcc per 16)", 64, 0)
    R(M_CND_SGPR, "v_cndmask_b32_e64 SGPR-pair mask", 64, 0)
    R(M_CMP_VCC, "v_cmp_le_f32 -> vcc", 64, 0)
    R(M_CMP_SGPR, "v_cmp_le_f32 -> SGPR pair", 64, 0)
    R(M_EXP, "v_exp_f32", 64, 0)
    R(M_EXP_NEG, "v_exp_f32_e64 with neg", 64, 0)
    R(M_RCP, "v_rcp_f32", 64, 0)
    R(M_MAX, "v_max_f32", 64, 0)
    R(M_MOV, "v_mov_b32", 64, 0)
    R(M_DPP_QUAD, "v_add_f32_dpp quad_perm", 64, 0)
    R(M_DPP_ROR, "v_add_f32_dpp row_ror:4", 64, 0)
    R(M_SUB_ABS, "v_sub_f32 |a|,|b|", 64, 0)
    R(M_FMA_NEG, "v_fma_f32 -a,b,b", 64, 0)
    R(M_PK_FMA, "v_pk_fma_f32 (2 FMAs per instruction)", 64, 0)
    R(M_PK_MUL, "v_pk_mul_f32", 64, 0)
    R(M_ADD_U32, "v_add_u32", 64, 0)
    R(M_AND, "v_and_b32", 64, 0)
    R(M_FMA_EXP_4_1, "3 v_fma + 1 v_exp (per instruction)", 64, 0)
    R(M_FMA_SALU, "v_fma with 1 s_add per 4 (per VALU inst)", 64, 0)
    R(M_MFMA_ONLY, "v_mfma_f32_16x16x4_f32 alone (per MFMA)", 0, 16)
    R(M_FMA_MFMA_16_1, "16 v_fma + 1 mfma16x16x4 (per VALU inst)", 64, 0)
    R(M_FMA_MFMA_8_1, "8 v_fma + 1 mfma16x16x4 (per VALU inst)", 64, 0)
    R(M_FMA_MFMA_4_1, "4 v_fma + 1 mfma16x16x4 (per VALU inst)", 64, 0)
    if (f) fclose(f);
    return 0;
}
