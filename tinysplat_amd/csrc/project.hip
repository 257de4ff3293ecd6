// project.hip - per-Gaussian streaming kernels for gfx950: EWA projection (fwd/bwd) and spherical
// harmonics (fwd/bwd).  One lane per Gaussian; all four are HBM-bound (SURVEY.md 8(d) D5:
// 96 N / 144 N / (24+12K) N bytes), so the work is in the access pattern, not the arithmetic:
//   * [N,3]/[N,4]/[N,6] AoS rows are read/written with per-lane 12/16/24-byte accesses whose
//     wave footprint is one contiguous 768 B..1.5 KiB span (every fetched line fully used);
//   * SH coefficient rows (12*K bytes per Gaussian, 192 B at degree 3) are moved with 16-byte
//     coalesced block transfers and transposed through LDS with an odd row stride, so the
//     per-lane row walk is bank-conflict free.
// This translation unit is compiled with -ffp-contract=off: project_fwd must reproduce the float32
// oracle bit-for-bit (radii / num_tiles_hit feed bit-exact binning checks).
#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"
#include "pack.h"
#include "splat_math.h"
#include "adam_math.h"

namespace {

constexpr int kThreads = 256;
// Gaussians (= threads) per workgroup of the fused colour stage.  Measured on config 3: forward 128 (23 KB of LDS
// per workgroup, six of them per CU: 72 -> 63 us with the non-temporal stores below), backward 256 (128: 51 -> 56 us).
#ifndef TS_SH_THREADS_FWD
#define TS_SH_THREADS_FWD 128
#endif
#ifndef TS_SH_THREADS_BWD
#define TS_SH_THREADS_BWD 256
#endif
#ifndef TS_NT_STORE
// Write-once gradient streams (192 N bytes of v_colors_rest per frame) leave with NON-TEMPORAL stores: they do
// not displace the coefficient rows that the next frame's forward reads from the 256 MB Infinity Cache
// (measured on config 3: sh_colors_bwd 51 -> 47 us and colors_pack_fwd 73 -> 66 us).  TS_NT_STORE=0 for A/B timing.
#define TS_NT_STORE 1
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4(float4* p, float a, float b, float c, float d) {
    if (TS_NT_STORE) {
        f4v v = {a, b, c, d};
        __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(p));
    } else {
        *p = make_float4(a, b, c, d);
    }
}
#ifndef TS_NT_GRADS
#define TS_NT_GRADS 0        // project_bwd's three gradient streams non-temporal (A/B knob)
#endif
constexpr int kShThreadsFwd = TS_SH_THREADS_FWD, kShThreadsBwd = TS_SH_THREADS_BWD;
#ifndef TS_NT_LOAD
// the 180-byte coefficient rows are read once per frame: non-temporal loads (colors_pack_fwd 64 -> 56 us on config 3)
#define TS_NT_LOAD 1
#endif
__device__ __forceinline__ float4 load4_stream(const float4* p) {
    if (TS_NT_LOAD) {
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

#include "project_inputs.h"

__global__ __launch_bounds__(kThreads) void project_fwd_kernel(
    int n, const float* __restrict__ means3d, const float* __restrict__ scales,
    const float* __restrict__ quats, const float* __restrict__ viewmat,
    const float* __restrict__ projmat, const ts_camera cam, const int flags,
    float* __restrict__ xys, float* __restrict__ depths, int* __restrict__ radii,
    float* __restrict__ conics, int* __restrict__ num_tiles_hit, float* __restrict__ cov3d) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const ts::Cam C = load_cam(viewmat, projmat, cam);
    const float m[3] = {means3d[3 * i], means3d[3 * i + 1], means3d[3 * i + 2]};
    float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    const float4 qv = reinterpret_cast<const float4*>(quats)[i];
    float q[4] = {qv.x, qv.y, qv.z, qv.w};
    prep_inputs(flags, s, q, nullptr);
    ts::ProjOut o;
    ts::project_one(C, m, s, q, o);
    reinterpret_cast<float2*>(xys)[i] = make_float2(o.x, o.y);
    depths[i] = o.depth;
    radii[i] = o.radius;
    conics[3 * i] = o.conic[0]; conics[3 * i + 1] = o.conic[1]; conics[3 * i + 2] = o.conic[2];
    num_tiles_hit[i] = o.tiles;
    if (cov3d) {
        float2* c3 = reinterpret_cast<float2*>(cov3d) + 3 * (size_t)i;
        c3[0] = make_float2(o.cov3d[0], o.cov3d[1]);
        c3[1] = make_float2(o.cov3d[2], o.cov3d[3]);
        c3[2] = make_float2(o.cov3d[4], o.cov3d[5]);
    }
}

// FUSED ADAM (round 6; SURVEY 8(f) F1: scripts/train.py:93-97, model_gaussian.py:112-120).  A training step used to
// write the six parameter gradients (236 bytes per Gaussian at SH degree 3) only for ts_adam_step to read them once.
// With ADAM the kernels that hold a gradient in registers apply the update there (adam_math.h: the very update of
// ts_adam_step, bit for bit): the projection's backward pass updates means / scales / quats - and the opacity logits,
// whose gradient ts_reduce_partials left in v_opacity -, the colour stage's backward pass colors_dc / colors_rest.
// No parameter gradient is materialised; xys.grad (what densification reads, model_gaussian.py:130-132) still is.
// A culled Gaussian has a zero gradient and is updated with it, as torch.optim.Adam does with a dense gradient.
struct ProjAdam {
    float *means, *scales, *quats, *opacities;            // the model's raw parameter tensors (log-scales, raw quaternions, logits)
    const float* v_opacity;                                // dL / d logit (ts_reduce_partials with TS_RASTER_LOGIT_OPACITY)
    float *m_means, *v_means, *m_scales, *v_scales, *m_quats, *v_quats, *m_opac, *v_opac;
    ts::AdamCoef c_means, c_scales, c_quats, c_opac;
};
template <bool ADAM>
__global__ __launch_bounds__(kThreads) void project_bwd_kernel(
    int n, const float* __restrict__ means3d, const float* __restrict__ scales,
    const float* __restrict__ quats, const float* __restrict__ viewmat,
    const float* __restrict__ projmat, const ts_camera cam, const int flags,
    const int* __restrict__ radii, const float* __restrict__ v_xy, const float* __restrict__ v_depth,
    const float* __restrict__ v_conic, const float* __restrict__ v_cov3d,
    float* __restrict__ v_means3d, float* __restrict__ v_scales, float* __restrict__ v_quats, const ProjAdam ad) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    ts::ProjGrad g;
    for (int k = 0; k < 3; ++k) { g.v_mean[k] = 0.0f; g.v_scale[k] = 0.0f; }
    for (int k = 0; k < 4; ++k) g.v_quat[k] = 0.0f;
    float pm[3] = {0.f, 0.f, 0.f}, ps[3] = {0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};      // (ADAM) the raw parameters
    if (ADAM) {
        pm[0] = means3d[3 * i]; pm[1] = means3d[3 * i + 1]; pm[2] = means3d[3 * i + 2];
        ps[0] = scales[3 * i]; ps[1] = scales[3 * i + 1]; ps[2] = scales[3 * i + 2];
        const float4 qv = reinterpret_cast<const float4*>(quats)[i];
        pq[0] = qv.x; pq[1] = qv.y; pq[2] = qv.z; pq[3] = qv.w;
    }
    if (radii[i] > 0) {
        const ts::Cam C = load_cam(viewmat, projmat, cam);
        const float m[3] = {ADAM ? pm[0] : means3d[3 * i], ADAM ? pm[1] : means3d[3 * i + 1], ADAM ? pm[2] : means3d[3 * i + 2]};
        float s[3] = {ADAM ? ps[0] : scales[3 * i], ADAM ? ps[1] : scales[3 * i + 1], ADAM ? ps[2] : scales[3 * i + 2]};
        float4 qv;
        if (ADAM) qv = make_float4(pq[0], pq[1], pq[2], pq[3]);
        else qv = reinterpret_cast<const float4*>(quats)[i];
        float q[4] = {qv.x, qv.y, qv.z, qv.w};
        float inv_n = 1.0f;
        prep_inputs(flags, s, q, &inv_n);
        const float2 vxy = reinterpret_cast<const float2*>(v_xy)[i];
        const float vx[2] = {vxy.x, vxy.y};
        const float vc[3] = {v_conic[3 * i], v_conic[3 * i + 1], v_conic[3 * i + 2]};
        float vcov[6];
        if (v_cov3d) {
            for (int k = 0; k < 6; ++k) vcov[k] = v_cov3d[6 * (size_t)i + k];
        }
        ts::project_one_vjp(C, m, s, q, vx, v_depth ? v_depth[i] : 0.0f, vc, v_cov3d ? vcov : nullptr, g);
        if (flags & TS_PROJECT_LOG_SCALES) {            // d exp(x) = exp(x) dx
            g.v_scale[0] *= s[0]; g.v_scale[1] *= s[1]; g.v_scale[2] *= s[2];
        }
        if (flags & TS_PROJECT_RAW_QUATS) {             // q_hat = q / |q|
            const float dotp = q[0] * g.v_quat[0] + q[1] * g.v_quat[1] + q[2] * g.v_quat[2] +
                               q[3] * g.v_quat[3];
            for (int k = 0; k < 4; ++k) g.v_quat[k] = (g.v_quat[k] - q[k] * dotp) * inv_n;
        }
    }
    if (ADAM) {
        float mo[3], vo[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { mo[k] = ad.m_means[3 * i + k]; vo[k] = ad.v_means[3 * i + k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ts::adam_update(pm[k], g.v_mean[k], mo[k], vo[k], ad.c_means);
            ad.means[3 * i + k] = pm[k]; ad.m_means[3 * i + k] = mo[k]; ad.v_means[3 * i + k] = vo[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { mo[k] = ad.m_scales[3 * i + k]; vo[k] = ad.v_scales[3 * i + k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ts::adam_update(ps[k], g.v_scale[k], mo[k], vo[k], ad.c_scales);
            ad.scales[3 * i + k] = ps[k]; ad.m_scales[3 * i + k] = mo[k]; ad.v_scales[3 * i + k] = vo[k];
        }
        float4 mq = reinterpret_cast<float4*>(ad.m_quats)[i], vq = reinterpret_cast<float4*>(ad.v_quats)[i];
        ts::adam_update(pq[0], g.v_quat[0], mq.x, vq.x, ad.c_quats); ts::adam_update(pq[1], g.v_quat[1], mq.y, vq.y, ad.c_quats);
        ts::adam_update(pq[2], g.v_quat[2], mq.z, vq.z, ad.c_quats); ts::adam_update(pq[3], g.v_quat[3], mq.w, vq.w, ad.c_quats);
        reinterpret_cast<float4*>(ad.quats)[i] = make_float4(pq[0], pq[1], pq[2], pq[3]);
        reinterpret_cast<float4*>(ad.m_quats)[i] = mq;
        reinterpret_cast<float4*>(ad.v_quats)[i] = vq;
        if (ad.opacities) {
            float po = ad.opacities[i], mo1 = ad.m_opac[i], vo1 = ad.v_opac[i];
            ts::adam_update(po, ad.v_opacity[i], mo1, vo1, ad.c_opac);
            ad.opacities[i] = po; ad.m_opac[i] = mo1; ad.v_opac[i] = vo1;
        }
        return;
    }
#if TS_NT_GRADS
    __builtin_nontemporal_store(g.v_mean[0], v_means3d + 3 * i); __builtin_nontemporal_store(g.v_mean[1], v_means3d + 3 * i + 1);
    __builtin_nontemporal_store(g.v_mean[2], v_means3d + 3 * i + 2);
    __builtin_nontemporal_store(g.v_scale[0], v_scales + 3 * i); __builtin_nontemporal_store(g.v_scale[1], v_scales + 3 * i + 1);
    __builtin_nontemporal_store(g.v_scale[2], v_scales + 3 * i + 2);
    store4(reinterpret_cast<float4*>(v_quats) + i, g.v_quat[0], g.v_quat[1], g.v_quat[2], g.v_quat[3]);
#else
    v_means3d[3 * i] = g.v_mean[0]; v_means3d[3 * i + 1] = g.v_mean[1];
    v_means3d[3 * i + 2] = g.v_mean[2];
    v_scales[3 * i] = g.v_scale[0]; v_scales[3 * i + 1] = g.v_scale[1];
    v_scales[3 * i + 2] = g.v_scale[2];
    reinterpret_cast<float4*>(v_quats)[i] =
        make_float4(g.v_quat[0], g.v_quat[1], g.v_quat[2], g.v_quat[3]);
#endif
}

// ------------------------------------------------------------------------------------------------
// SH.  A block owns 256 consecutive Gaussians = one contiguous span of 256*3K floats of `coeffs`.
// LDS row stride RSP = 3*Ka rounded up to an odd number of floats: lane t walks row t with
// ds_read_b32 at (t*RSP + j) -> 32 distinct banks per half-wave.
// ------------------------------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(kThreads) void sh_fwd_kernel(
    int n, int num_bases, const float* __restrict__ viewdirs, const float* __restrict__ coeffs,
    float* __restrict__ colors) {
    constexpr int KA = (DEG + 1) * (DEG + 1);       // active bases
    constexpr int RS = 3 * KA;                      // active floats per row
    constexpr int RSP = RS | 1;                     // odd LDS stride
    extern __shared__ __align__(16) float lds[];
    const int g0 = blockIdx.x * kThreads;
    const int cnt = min(kThreads, n - g0);
    const int tid = threadIdx.x;
    const size_t row = 3 * (size_t)num_bases;       // stored floats per row
    const float* src = coeffs + (size_t)g0 * row;
    if (num_bases == KA && cnt == kThreads) {
        // full rows, full block: 16-byte coalesced transfers of the whole span
        const float4* src4 = reinterpret_cast<const float4*>(src);
        constexpr int total4 = kThreads * RS / 4;
        for (int f4 = tid; f4 < total4; f4 += kThreads) {
            const float4 v = src4[f4];
            const int f = 4 * f4;
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ff = f + u;
                lds[(ff / RS) * RSP + (ff % RS)] = e[u];
            }
        }
    } else {
        const int total = cnt * RS;
        for (int f = tid; f < total; f += kThreads) {
            const int g = f / RS, j = f % RS;
            lds[g * RSP + j] = src[(size_t)g * row + j];
        }
    }
    __syncthreads();
    if (tid >= cnt) return;
    const int i = g0 + tid;
    float Y[KA];
    ts::sh_basis(DEG, viewdirs[3 * i], viewdirs[3 * i + 1], viewdirs[3 * i + 2], Y);
    const float* r = lds + tid * RSP;
    float c0 = Y[0] * r[0], c1 = Y[0] * r[1], c2 = Y[0] * r[2];
#pragma unroll
    for (int k = 1; k < KA; ++k) {
        c0 = c0 + Y[k] * r[3 * k];
        c1 = c1 + Y[k] * r[3 * k + 1];
        c2 = c2 + Y[k] * r[3 * k + 2];
    }
    colors[3 * i] = c0; colors[3 * i + 1] = c1; colors[3 * i + 2] = c2;
}

template <int DEG>
__global__ __launch_bounds__(kThreads) void sh_bwd_kernel(
    int n, int num_bases, const float* __restrict__ viewdirs, const float* __restrict__ v_colors,
    float* __restrict__ v_coeffs) {
    constexpr int KA = (DEG + 1) * (DEG + 1);
    extern __shared__ __align__(16) float lds[];
    const int RS = 3 * num_bases;                   // full stored row (inactive bands get zeros)
    const int RSP = RS | 1;
    const int g0 = blockIdx.x * kThreads;
    const int cnt = min(kThreads, n - g0);
    const int tid = threadIdx.x;
    if (tid < cnt) {
        const int i = g0 + tid;
        float Y[KA];
        ts::sh_basis(DEG, viewdirs[3 * i], viewdirs[3 * i + 1], viewdirs[3 * i + 2], Y);
        const float v0 = v_colors[3 * i], v1 = v_colors[3 * i + 1], v2 = v_colors[3 * i + 2];
        float* r = lds + tid * RSP;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            r[3 * k] = Y[k] * v0; r[3 * k + 1] = Y[k] * v1; r[3 * k + 2] = Y[k] * v2;
        }
        for (int j = 3 * KA; j < RS; ++j) r[j] = 0.0f;
    }
    __syncthreads();
    float* dst = v_coeffs + (size_t)g0 * RS;
    const int total = cnt * RS;
    if ((total & 3) == 0) {
        float4* dst4 = reinterpret_cast<float4*>(dst);    // g0*RS*4 bytes is a multiple of 16
        for (int f4 = tid; f4 < total / 4; f4 += kThreads) {
            float e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ff = 4 * f4 + u;
                e[u] = lds[(ff / RS) * RSP + (ff % RS)];
            }
            dst4[f4] = make_float4(e[0], e[1], e[2], e[3]);
        }
    } else {
        for (int f = tid; f < total; f += kThreads) dst[f] = lds[(f / RS) * RSP + (f % RS)];
    }
}

// ------------------------------------------------------------------------------------------------
// Fused colour stage of the adapter (rasterize.py:75-81 + :38-39): view directions from the means
// and the view-matrix translation column, SH evaluation on the SPLIT coefficient tensors
// colors_dc[N,3] / colors_rest[N,K-1,3] (no 192 N-byte torch.cat), then clamp(rgb + 0.5, min=0).
// mask[n] bit c is set where the clamp passes gradient (pre-clamp value >= 0, torch's rule).
// ------------------------------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(kShThreadsFwd) void sh_colors_fwd_kernel(
    int n, int num_bases, const float* __restrict__ means, const float* __restrict__ origin,
    const float* __restrict__ dc, const float* __restrict__ rest, float* __restrict__ colors,
    unsigned char* __restrict__ mask, const ts::PackArgs pk) {
    constexpr int KA = (DEG + 1) * (DEG + 1);
    constexpr int RS = 3 * (KA - 1);                // active floats of a `rest` row
    constexpr int RSP = RS | 1;
    extern __shared__ __align__(16) float lds[];
    const int g0 = blockIdx.x * kShThreadsFwd;
    const int cnt = min(kShThreadsFwd, n - g0);
    const int tid = threadIdx.x;
    if (RS > 0) {
        const size_t row = 3 * (size_t)(num_bases - 1);
        const float* src = rest + (size_t)g0 * row;
        if (num_bases == KA && cnt == kShThreadsFwd && ((kShThreadsFwd * RS) & 3) == 0) {
            // all 16-byte loads of the span are issued before the first LDS write, so every lane
            // has its ~RS/4 requests in flight at once (the kernel is a pure HBM stream)
            const float4* src4 = reinterpret_cast<const float4*>(src);
            constexpr int total4 = kShThreadsFwd * RS / 4;
            constexpr int per = (total4 + kShThreadsFwd - 1) / kShThreadsFwd;
            float4 v[per > 0 ? per : 1];
#pragma unroll
            for (int u = 0; u < per; ++u) {
                const int f4 = tid + u * kShThreadsFwd;
                v[u] = f4 < total4 ? load4_stream(src4 + f4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < per; ++u) {
                const int f4 = tid + u * kShThreadsFwd;
                if (f4 < total4) {
                    const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int ff = 4 * f4 + c;
                        lds[(ff / RS) * RSP + (ff % RS)] = e[c];
                    }
                }
            }
        } else {
            const int total = cnt * RS;
            for (int f = tid; f < total; f += kShThreadsFwd) {
                const int g = f / RS, j = f % RS;
                lds[g * RSP + j] = src[(size_t)g * row + j];
            }
        }
        __syncthreads();
    }
    if (tid >= cnt) return;
    const int i = g0 + tid;
    float Y[KA];
    ts::sh_basis(DEG, means[3 * i] - origin[0], means[3 * i + 1] - origin[1],
                 means[3 * i + 2] - origin[2], Y);
    float c0 = Y[0] * dc[3 * i], c1 = Y[0] * dc[3 * i + 1], c2 = Y[0] * dc[3 * i + 2];
    const float* r = lds + tid * RSP;
#pragma unroll
    for (int k = 1; k < KA; ++k) {
        c0 = c0 + Y[k] * r[3 * (k - 1)];
        c1 = c1 + Y[k] * r[3 * (k - 1) + 1];
        c2 = c2 + Y[k] * r[3 * (k - 1) + 2];
    }
    c0 = c0 + 0.5f; c1 = c1 + 0.5f; c2 = c2 + 0.5f;
    if (mask) mask[i] = (unsigned char)((c0 >= 0.0f ? 1 : 0) | (c1 >= 0.0f ? 2 : 0) | (c2 >= 0.0f ? 4 : 0));
    c0 = fmaxf(c0, 0.0f); c1 = fmaxf(c1, 0.0f); c2 = fmaxf(c2, 0.0f);
    if (colors) { colors[3 * i] = c0; colors[3 * i + 1] = c1; colors[3 * i + 2] = c2; }
    if (pk.splats) ts::pack_one(pk, i, c0, c1, c2, pk.channels == 4 ? pk.depths[i] : 0.0f);
}

// The same colour stage for a tile-row STRIPE of a multi-GPU frame: only the Gaussians that are listed in
// the stripe (live[i] = num_tiles_hit[i] != 0, typically 1/G of them) are evaluated; each live lane reads
// its own coefficient row (no LDS staging: a workgroup holds few live rows, and skipping the others is the
// point - the dense kernel streams all 12 K bytes per Gaussian).  Colours and masks of the other
// Gaussians are left unwritten: nothing reads them (ts_pack_splats and ts_reduce_partials touch only
// listed Gaussians).
template <int DEG>
__global__ __launch_bounds__(kThreads) void sh_colors_fwd_sparse_kernel(
    int n, int num_bases, const int* __restrict__ live, const float* __restrict__ means,
    const float* __restrict__ origin, const float* __restrict__ dc, const float* __restrict__ rest,
    float* __restrict__ colors, unsigned char* __restrict__ mask, const ts::PackArgs pk) {
    constexpr int KA = (DEG + 1) * (DEG + 1);
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n || live[i] == 0) return;
    float Y[KA];
    ts::sh_basis(DEG, means[3 * i] - origin[0], means[3 * i + 1] - origin[1],
                 means[3 * i + 2] - origin[2], Y);
    float c0 = Y[0] * dc[3 * i], c1 = Y[0] * dc[3 * i + 1], c2 = Y[0] * dc[3 * i + 2];
    const float* r = rest + (size_t)i * 3 * (num_bases - 1);
#pragma unroll
    for (int k = 1; k < KA; ++k) {      // same association order as the dense kernel: identical bits
        c0 = c0 + Y[k] * r[3 * (k - 1)];
        c1 = c1 + Y[k] * r[3 * (k - 1) + 1];
        c2 = c2 + Y[k] * r[3 * (k - 1) + 2];
    }
    c0 = c0 + 0.5f; c1 = c1 + 0.5f; c2 = c2 + 0.5f;
    if (mask) mask[i] = (unsigned char)((c0 >= 0.0f ? 1 : 0) | (c1 >= 0.0f ? 2 : 0) | (c2 >= 0.0f ? 4 : 0));
    c0 = fmaxf(c0, 0.0f); c1 = fmaxf(c1, 0.0f); c2 = fmaxf(c2, 0.0f);
    if (colors) { colors[3 * i] = c0; colors[3 * i + 1] = c1; colors[3 * i + 2] = c2; }
    if (pk.splats) ts::pack_one(pk, i, c0, c1, c2, pk.channels == 4 ? pk.depths[i] : 0.0f);
}

struct ShAdam {             // FUSED ADAM (see project_bwd_kernel): colors_dc / colors_rest updated from the gradient rows in registers
    float *dc, *rest, *m_dc, *v_dc, *m_rest, *v_rest;
    ts::AdamCoef c_dc, c_rest;
};
template <int DEG, bool ADAM>
__global__ __launch_bounds__(kShThreadsBwd) void sh_colors_bwd_kernel(
    int n, int num_bases, const float* __restrict__ means, const float* __restrict__ origin,
    const unsigned char* __restrict__ mask, const float* __restrict__ v_colors,
    float* __restrict__ v_dc, float* __restrict__ v_rest, const ShAdam ad) {
    constexpr int KA = (DEG + 1) * (DEG + 1);
    extern __shared__ __align__(16) float lds[];
    const int RS = 3 * (num_bases - 1);
    const int RSP = RS | 1;
    const int g0 = blockIdx.x * kShThreadsBwd;
    const int cnt = min(kShThreadsBwd, n - g0);
    const int tid = threadIdx.x;
    if (tid < cnt) {
        const int i = g0 + tid;
        float Y[KA];
        ts::sh_basis(DEG, means[3 * i] - origin[0], means[3 * i + 1] - origin[1],
                     means[3 * i + 2] - origin[2], Y);
        const int m = mask ? mask[i] : 7;      // NULL: the clamp was applied upstream (ts_reduce_partials)
        const float v0 = (m & 1) ? v_colors[3 * i] : 0.0f, v1 = (m & 2) ? v_colors[3 * i + 1] : 0.0f,
                    v2 = (m & 4) ? v_colors[3 * i + 2] : 0.0f;
        if (ADAM) {
            const float gd[3] = {Y[0] * v0, Y[0] * v1, Y[0] * v2};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float p_ = ad.dc[3 * i + c], m_ = ad.m_dc[3 * i + c], s_ = ad.v_dc[3 * i + c];
                ts::adam_update(p_, gd[c], m_, s_, ad.c_dc);
                ad.dc[3 * i + c] = p_; ad.m_dc[3 * i + c] = m_; ad.v_dc[3 * i + c] = s_;
            }
        } else {
            v_dc[3 * i] = Y[0] * v0; v_dc[3 * i + 1] = Y[0] * v1; v_dc[3 * i + 2] = Y[0] * v2;
        }
        float* r = lds + tid * RSP;
#pragma unroll
        for (int k = 1; k < KA; ++k) {
            r[3 * (k - 1)] = Y[k] * v0; r[3 * (k - 1) + 1] = Y[k] * v1; r[3 * (k - 1) + 2] = Y[k] * v2;
        }
        for (int j = 3 * (KA - 1); j < RS; ++j) r[j] = 0.0f;
    }
    if (RS == 0) return;
    __syncthreads();
    const size_t off = (size_t)g0 * RS;
    float* dst = ADAM ? nullptr : v_rest + off;
    const int total = cnt * RS;
    if ((total & 3) == 0 && ((off) & 3) == 0) {
        float4* dst4 = reinterpret_cast<float4*>(dst);
        for (int f4 = tid; f4 < total / 4; f4 += kShThreadsBwd) {
            float e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ff = 4 * f4 + u;
                e[u] = lds[(ff / RS) * RSP + (ff % RS)];
            }
            if (ADAM) {       // the same 16-byte chunk of the parameter and of its two moments: updated where the gradient is
                float4* p4 = reinterpret_cast<float4*>(ad.rest + off) + f4;
                float4* m4 = reinterpret_cast<float4*>(ad.m_rest + off) + f4;
                float4* s4 = reinterpret_cast<float4*>(ad.v_rest + off) + f4;
                float4 pp = *p4, mm = *m4, ss = *s4;
                ts::adam_update(pp.x, e[0], mm.x, ss.x, ad.c_rest); ts::adam_update(pp.y, e[1], mm.y, ss.y, ad.c_rest);
                ts::adam_update(pp.z, e[2], mm.z, ss.z, ad.c_rest); ts::adam_update(pp.w, e[3], mm.w, ss.w, ad.c_rest);
                *p4 = pp; *m4 = mm; *s4 = ss;
            } else {
                store4(dst4 + f4, e[0], e[1], e[2], e[3]);
            }
        }
    } else {
        for (int f = tid; f < total; f += kShThreadsBwd) {
            const float e = lds[(f / RS) * RSP + (f % RS)];
            if (ADAM) {
                float p_ = ad.rest[off + f], m_ = ad.m_rest[off + f], s_ = ad.v_rest[off + f];
                ts::adam_update(p_, e, m_, s_, ad.c_rest);
                ad.rest[off + f] = p_; ad.m_rest[off + f] = m_; ad.v_rest[off + f] = s_;
            } else {
                dst[f] = e;
            }
        }
    }
}

// Measurement utility (bench.py): streaming read of n16 16-byte words with a
// grid-stride loop, 8 independent loads in flight per lane; the per-lane sums are folded into
// sink[] so that the loads cannot be elided.  Gives the achievable HBM read bandwidth of this GPU.
__global__ __launch_bounds__(kThreads) void stream_read_kernel(const float4* __restrict__ src,
                                                               size_t n16, float* __restrict__ sink) {
    constexpr int U = 8;       // 8 x 16 B in flight per lane: 6.3 TB/s on MI355X (4: 5.6, tools/micro/stream_bench.hip)
    const size_t stride = (size_t)gridDim.x * kThreads;
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    float4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {         // read-once stream: non-temporal (6.8 vs 6.3 TB/s in stream_bench.hip)
            const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(src + i + u * stride));
            v[u] = make_float4(t.x, t.y, t.z, t.w);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w;
        }
    }
    for (; i < n16; i += stride) {
        const float4 v0 = src[i];
        acc[0].x += v0.x; acc[0].y += v0.y; acc[0].z += v0.z; acc[0].w += v0.w;
    }
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) t += (acc[u].x + acc[u].y) + (acc[u].z + acc[u].w);
    if (t == 123456.789f) sink[0] = t;          // practically never true: keeps the loads alive
}

// Measurement utility (bench.py): the access pattern of the compositing kernels' record gather - lane j reads the
// 48-byte record ids[j] with three 16-byte loads - on a KNOWN number of records, so that rocprofv3's FETCH_SIZE
// can be calibrated for this pattern (MI355X_MICROARCH.md, HBM section: only wide coalesced reads are calibrated).
__global__ __launch_bounds__(kThreads) void gather48_kernel(const float4* __restrict__ records,
                                                            const int* __restrict__ ids, size_t m,
                                                            float* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t j = (size_t)blockIdx.x * kThreads + threadIdx.x; j < m; j += stride) {
        const size_t g = (size_t)ids[j];
        const float4 q0 = records[3 * g], q1 = records[3 * g + 1], q2 = records[3 * g + 2];
        a.x += q0.x + q1.x + q2.x; a.y += q0.y + q1.y + q2.y; a.z += q0.z + q1.z + q2.z; a.w += q0.w + q1.w + q2.w;
    }
    if ((a.x + a.y) + (a.z + a.w) == 123456.789f) sink[0] = a.x;
}

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int ts_abi_version(void) { return TS_ABI_VERSION; }

int ts_project_fwd(int32_t n, const float* means3d, const float* scales, const float* quats,
                   const float* viewmat, const float* projmat, const ts_camera* cam,
                   int32_t flags, float* xys, float* depths, int32_t* radii, float* conics,
                   int32_t* num_tiles_hit, float* cov3d, void* stream) {
    if (n < 0 || !cam) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!means3d || !scales || !quats || !viewmat || !projmat || !xys || !depths || !radii ||
        !conics || !num_tiles_hit)
        return TS_E_BADARG;
    const int grid = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(project_fwd_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, n,
                       means3d, scales, quats, viewmat, projmat, *cam, (int)flags, xys, depths, radii,
                       conics, num_tiles_hit, cov3d);
    return launch_status();
}

int ts_project_bwd(int32_t n, const float* means3d, const float* scales, const float* quats,
                   const float* viewmat, const float* projmat, const ts_camera* cam,
                   int32_t flags, const int32_t* radii, const float* v_xy, const float* v_depth,
                   const float* v_conic, const float* v_cov3d, float* v_means3d, float* v_scales,
                   float* v_quats, void* stream) {
    if (n < 0 || !cam) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!means3d || !scales || !quats || !viewmat || !projmat || !radii || !v_xy ||
        !v_conic || !v_means3d || !v_scales || !v_quats)
        return TS_E_BADARG;
    const int grid = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(project_bwd_kernel<false>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, n,
                       means3d, scales, quats, viewmat, projmat, *cam, (int)flags, radii, v_xy, v_depth,
                       v_conic, v_cov3d, v_means3d, v_scales, v_quats, ProjAdam{});
    return launch_status();
}

static bool adam_group_ok(const ts_adam* a, int k) {
    return a->exp_avg[k] && a->exp_avg_sq[k] && a->step[k] >= 1;
}

int ts_project_bwd_adam(int32_t n, float* means3d, float* scales, float* quats, const float* viewmat,
                        const float* projmat, const ts_camera* cam, int32_t flags, const int32_t* radii,
                        const float* v_xy, const float* v_depth, const float* v_conic, float* opacities,
                        const float* v_opacity, const ts_adam* adam, void* stream) {
    if (n < 0 || !cam || !adam) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!means3d || !scales || !quats || !viewmat || !projmat || !radii || !v_xy || !v_conic) return TS_E_BADARG;
    if (!adam_group_ok(adam, TS_ADAM_MEANS) || !adam_group_ok(adam, TS_ADAM_SCALES) || !adam_group_ok(adam, TS_ADAM_QUATS))
        return TS_E_BADARG;
    if (opacities && (!v_opacity || !adam_group_ok(adam, TS_ADAM_OPACITIES))) return TS_E_BADARG;
    ProjAdam ad{};
    ad.means = means3d; ad.scales = scales; ad.quats = quats; ad.opacities = opacities; ad.v_opacity = v_opacity;
    ad.m_means = adam->exp_avg[TS_ADAM_MEANS]; ad.v_means = adam->exp_avg_sq[TS_ADAM_MEANS];
    ad.m_scales = adam->exp_avg[TS_ADAM_SCALES]; ad.v_scales = adam->exp_avg_sq[TS_ADAM_SCALES];
    ad.m_quats = adam->exp_avg[TS_ADAM_QUATS]; ad.v_quats = adam->exp_avg_sq[TS_ADAM_QUATS];
    ad.m_opac = adam->exp_avg[TS_ADAM_OPACITIES]; ad.v_opac = adam->exp_avg_sq[TS_ADAM_OPACITIES];
    auto coef = [&](int k) { return ts::adam_coef(adam->lr[k], adam->step[k], adam->beta1, adam->beta2, adam->eps); };
    ad.c_means = coef(TS_ADAM_MEANS); ad.c_scales = coef(TS_ADAM_SCALES); ad.c_quats = coef(TS_ADAM_QUATS);
    if (opacities) ad.c_opac = coef(TS_ADAM_OPACITIES);
    const int grid = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(project_bwd_kernel<true>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, n,
                       means3d, scales, quats, viewmat, projmat, *cam, (int)flags, radii, v_xy, v_depth,
                       v_conic, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, ad);
    return launch_status();
}

static int sh_check(int32_t n, int32_t degree, int32_t num_bases) {
    if (n < 0) return TS_E_BADARG;
    if (num_bases != 1 && num_bases != 4 && num_bases != 9 && num_bases != 16 && num_bases != 25)
        return TS_E_DEGREE;
    if (degree < 0 || degree > 4 || (degree + 1) * (degree + 1) > num_bases) return TS_E_DEGREE;
    return 0;
}

int ts_sh_fwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* viewdirs,
              const float* coeffs, float* colors, void* stream) {
    const int chk = sh_check(n, degrees_to_use, num_bases);
    if (chk) return chk;
    if (n == 0) return 0;
    if (!viewdirs || !coeffs || !colors) return TS_E_BADARG;
    const int grid = (n + kThreads - 1) / kThreads;
    const int ka = (degrees_to_use + 1) * (degrees_to_use + 1);
    const size_t lds = (size_t)kThreads * ((3 * ka) | 1) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define TS_SH_FWD(D)                                                                          \
    hipLaunchKernelGGL(sh_fwd_kernel<D>, dim3(grid), dim3(kThreads), lds, s, n, num_bases,     \
                       viewdirs, coeffs, colors)
    switch (degrees_to_use) {
        case 0: TS_SH_FWD(0); break;
        case 1: TS_SH_FWD(1); break;
        case 2: TS_SH_FWD(2); break;
        case 3: TS_SH_FWD(3); break;
        default: {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sh_fwd_kernel<4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            TS_SH_FWD(4);
        } break;
    }
#undef TS_SH_FWD
    return launch_status();
}

int ts_sh_bwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* viewdirs,
              const float* v_colors, float* v_coeffs, void* stream) {
    const int chk = sh_check(n, degrees_to_use, num_bases);
    if (chk) return chk;
    if (n == 0) return 0;
    if (!viewdirs || !v_colors || !v_coeffs) return TS_E_BADARG;
    const int grid = (n + kThreads - 1) / kThreads;
    const size_t lds = (size_t)kThreads * ((3 * num_bases) | 1) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define TS_SH_BWD(D)                                                                          \
    do {                                                                                      \
        if (lds > 48 * 1024)                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sh_bwd_kernel<D>),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        hipLaunchKernelGGL(sh_bwd_kernel<D>, dim3(grid), dim3(kThreads), lds, s, n, num_bases, \
                           viewdirs, v_colors, v_coeffs);                                     \
    } while (0)
    switch (degrees_to_use) {
        case 0: TS_SH_BWD(0); break;
        case 1: TS_SH_BWD(1); break;
        case 2: TS_SH_BWD(2); break;
        case 3: TS_SH_BWD(3); break;
        default: TS_SH_BWD(4); break;
    }
#undef TS_SH_BWD
    return launch_status();
}

static int launch_colors_fwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                             const float* origin, const float* colors_dc, const float* colors_rest,
                             float* colors, uint8_t* clamp_mask, const int32_t* live, const ts::PackArgs& pk,
                             void* stream) {
    const int chk = sh_check(n, degrees_to_use, num_bases);
    if (chk) return chk;
    if (n == 0) return 0;
    if (!means3d || !origin || !colors_dc || (!colors && !pk.splats) || (num_bases > 1 && !colors_rest))
        return TS_E_BADARG;
    const int grid = (n + kThreads - 1) / kThreads;
    hipStream_t s = (hipStream_t)stream;
    if (live) {
#define TS_SHC_SPARSE(D)                                                                           \
        hipLaunchKernelGGL(sh_colors_fwd_sparse_kernel<D>, dim3(grid), dim3(kThreads), 0, s, n,    \
                           num_bases, live, means3d, origin, colors_dc, colors_rest, colors, clamp_mask, pk)
        switch (degrees_to_use) {
            case 0: TS_SHC_SPARSE(0); break;
            case 1: TS_SHC_SPARSE(1); break;
            case 2: TS_SHC_SPARSE(2); break;
            case 3: TS_SHC_SPARSE(3); break;
            default: TS_SHC_SPARSE(4); break;
        }
#undef TS_SHC_SPARSE
        return launch_status();
    }
    const int ka = (degrees_to_use + 1) * (degrees_to_use + 1);
    const size_t lds = (size_t)kShThreadsFwd * ((3 * (ka - 1)) | 1) * sizeof(float);
    const int grid_d = (n + kShThreadsFwd - 1) / kShThreadsFwd;
#define TS_SHC_FWD(D)                                                                             \
    do {                                                                                          \
        if (lds > 48 * 1024)                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sh_colors_fwd_kernel<D>),    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        hipLaunchKernelGGL(sh_colors_fwd_kernel<D>, dim3(grid_d), dim3(kShThreadsFwd), lds, s, n,  \
                           num_bases, means3d, origin, colors_dc, colors_rest, colors, clamp_mask, pk); \
    } while (0)
    switch (degrees_to_use) {
        case 0: TS_SHC_FWD(0); break;
        case 1: TS_SHC_FWD(1); break;
        case 2: TS_SHC_FWD(2); break;
        case 3: TS_SHC_FWD(3); break;
        default: TS_SHC_FWD(4); break;
    }
#undef TS_SHC_FWD
    return launch_status();
}

int ts_sh_colors_fwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                     const float* origin, const float* colors_dc, const float* colors_rest,
                     float* colors, uint8_t* clamp_mask, const int32_t* live, void* stream) {
    ts::PackArgs none;
    none.splats = nullptr;
    if (!colors) return TS_E_BADARG;
    return launch_colors_fwd(n, degrees_to_use, num_bases, means3d, origin, colors_dc, colors_rest, colors,
                             clamp_mask, live, none, stream);
}

int ts_colors_pack_fwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                       const float* origin, const float* colors_dc, const float* colors_rest,
                       uint8_t* clamp_mask, const int32_t* live, int32_t channels, int32_t flags,
                       const float* xys, const int32_t* radii, const float* conics, const float* opacity,
                       const int32_t* cum_tiles_hit, const ts_camera* cam, const float* depths,
                       float* splats, void* stream) {
    if (n < 0 || !cam || (channels != 3 && channels != 4)) return TS_E_BADARG;
    if (n > 0 && (!xys || !radii || !conics || !opacity || !cum_tiles_hit || !splats ||
                  (channels == 4 && !depths)))
        return TS_E_BADARG;
    ts::PackArgs pk;
    pk.channels = channels; pk.flags = (int)flags; pk.xys = xys; pk.radii = radii; pk.conics = conics;
    pk.opacity = opacity; pk.cum_tiles_hit = cum_tiles_hit; pk.depths = depths;
    pk.splats = reinterpret_cast<float4*>(splats); pk.cam = *cam;
    return launch_colors_fwd(n, degrees_to_use, num_bases, means3d, origin, colors_dc, colors_rest, nullptr,
                             clamp_mask, live, pk, stream);
}

static int sh_colors_bwd_launch(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                                const float* origin, const uint8_t* clamp_mask, const float* v_colors,
                                float* v_colors_dc, float* v_colors_rest, const ShAdam* ad, void* stream) {
    const int grid = (n + kShThreadsBwd - 1) / kShThreadsBwd;
    const size_t lds = (size_t)kShThreadsBwd * ((3 * (num_bases - 1)) | 1) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define TS_SHC_BWD2(D, A)                                                                            \
    do {                                                                                             \
        if (lds > 48 * 1024)                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sh_colors_bwd_kernel<D, A>),    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL((sh_colors_bwd_kernel<D, A>), dim3(grid), dim3(kShThreadsBwd), lds, s, n, \
                           num_bases, means3d, origin, clamp_mask, v_colors, v_colors_dc,            \
                           v_colors_rest, ad ? *ad : ShAdam{});                                      \
    } while (0)
#define TS_SHC_BWD(D)                                       \
    do {                                                    \
        if (ad) TS_SHC_BWD2(D, true);                       \
        else TS_SHC_BWD2(D, false);                         \
    } while (0)
    switch (degrees_to_use) {
        case 0: TS_SHC_BWD(0); break;
        case 1: TS_SHC_BWD(1); break;
        case 2: TS_SHC_BWD(2); break;
        case 3: TS_SHC_BWD(3); break;
        default: TS_SHC_BWD(4); break;
    }
#undef TS_SHC_BWD
#undef TS_SHC_BWD2
    return launch_status();
}

int ts_sh_colors_bwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                     const float* origin, const uint8_t* clamp_mask, const float* v_colors,
                     float* v_colors_dc, float* v_colors_rest, void* stream) {
    const int chk = sh_check(n, degrees_to_use, num_bases);
    if (chk) return chk;
    if (n == 0) return 0;
    if (!means3d || !origin || !v_colors || !v_colors_dc || (num_bases > 1 && !v_colors_rest))
        return TS_E_BADARG;
    return sh_colors_bwd_launch(n, degrees_to_use, num_bases, means3d, origin, clamp_mask, v_colors, v_colors_dc,
                                v_colors_rest, nullptr, stream);
}

int ts_sh_colors_bwd_adam(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                          const float* origin, const uint8_t* clamp_mask, const float* v_colors,
                          float* colors_dc, float* colors_rest, const ts_adam* adam, void* stream) {
    const int chk = sh_check(n, degrees_to_use, num_bases);
    if (chk) return chk;
    if (n == 0) return 0;
    if (!adam || !means3d || !origin || !v_colors || !colors_dc || (num_bases > 1 && !colors_rest)) return TS_E_BADARG;
    if (!adam_group_ok(adam, TS_ADAM_COLORS_DC) || (num_bases > 1 && !adam_group_ok(adam, TS_ADAM_COLORS_REST)))
        return TS_E_BADARG;
    ShAdam ad{};
    ad.dc = colors_dc; ad.rest = colors_rest;
    ad.m_dc = adam->exp_avg[TS_ADAM_COLORS_DC]; ad.v_dc = adam->exp_avg_sq[TS_ADAM_COLORS_DC];
    ad.m_rest = adam->exp_avg[TS_ADAM_COLORS_REST]; ad.v_rest = adam->exp_avg_sq[TS_ADAM_COLORS_REST];
    ad.c_dc = ts::adam_coef(adam->lr[TS_ADAM_COLORS_DC], adam->step[TS_ADAM_COLORS_DC], adam->beta1, adam->beta2, adam->eps);
    if (num_bases > 1)
        ad.c_rest = ts::adam_coef(adam->lr[TS_ADAM_COLORS_REST], adam->step[TS_ADAM_COLORS_REST], adam->beta1, adam->beta2,
                                  adam->eps);
    return sh_colors_bwd_launch(n, degrees_to_use, num_bases, means3d, origin, clamp_mask, v_colors, nullptr, nullptr, &ad,
                                stream);
}

int ts_bench_gather48(const float* records, const int32_t* ids, int64_t m, float* sink, void* stream) {
    if (!records || !ids || !sink || m < 1) return TS_E_BADARG;
    hipLaunchKernelGGL(gather48_kernel, dim3(256 * 8), dim3(kThreads), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(records), ids, (size_t)m, sink);
    return launch_status();
}

int ts_bench_stream_read(const float* src, int64_t n_floats, float* sink, void* stream) {
    if (!src || !sink || n_floats < 4) return TS_E_BADARG;
    const size_t n16 = (size_t)n_floats / 4;
    hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 8), dim3(kThreads), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), n16, sink);
    return launch_status();
}

}  // extern "C"
