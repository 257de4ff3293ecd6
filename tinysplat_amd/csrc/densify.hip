// densify.hip - tinysplat's densification hooks as stream-compaction kernels.
//
// Reference behaviour (/root/reference/tinysplat/splatting/model_gaussian.py):
//   update_grad_accum   :130-132   accum += ||xys.grad||
//   densify_and_prune   :138-195   clone small / split large Gaussians with a large mean 2-D gradient,
//                                  prune faint-and-huge ones and the originals of the split ones
//   update_state        :197-242   rebuild the six parameter tensors and both Adam moments:
//                                  surviving rows in order, then the new rows (moments zero)
//   GaussianDistribution.sample :533-572   two samples per split Gaussian, scales / 1.6
//
// The reference does this with ~60 boolean-index / cat / repeat PyTorch ops, each a pass over HBM.
// Here: one elementwise classify pass -> one counting pass + single-block scan -> one "plan" pass that
// writes a row map  src_of[N']  (rows: kept | cloned | split sample 0 | split sample 1, each in source
// order - exactly the reference's concatenation order), after which every tensor is rebuilt by ONE
// gather launch per table (parameters; exp_avg; exp_avg_sq) and the 2S sampled rows get their means /
// scales from a small fix-up kernel.  All kernels are HBM streaming; the only host read is the four
// counts that size the new tensors.  No atomics anywhere.
//
// Built with -ffp-contract=off: the threshold comparisons follow the reference's operation order.
#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 4;                         // Gaussians per thread in the count / plan passes
constexpr int kBlockItems = kThreads * kItems;    // 1024 per block: each of the three counters needs 11 bits,
constexpr int kFieldBits = 21;                    // so they are scanned together packed in one 64-bit word
constexpr long long kFieldMask = (1ll << kFieldBits) - 1;
typedef long long packed_t;

inline int launch_status() { return (int)hipGetLastError(); }
inline int num_blocks(int n) { return (n + kBlockItems - 1) / kBlockItems; }

__global__ __launch_bounds__(kThreads) void grad_accum_kernel(int n, const float2* __restrict__ v_xy,
                                                              float* __restrict__ accum) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float2 g = v_xy[i];
    // the reference's `xys.grad.norm(dim=-1)` (torch's reduction: acc = fma(v, v, acc) over the two
    // components, then a correctly rounded float32 root) - reproduced operation for operation
    accum[i] += sqrtf(fmaf(g.y, g.y, g.x * g.x));
}

__global__ __launch_bounds__(kThreads) void classify_kernel(int n, const float* __restrict__ accum,
                                                            const float* __restrict__ scales,
                                                            const float* __restrict__ opacities,
                                                            ts_densify_policy p,
                                                            uint8_t* __restrict__ flags) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    // :148  grad_norm_avg = accum / interval / 2 * max(width, height), left to right
    const float g = accum[i] / p.interval_densify / 2.0f * p.max_dim;
    const bool grad = g >= p.tau_means;
    const float ms = fmaxf(fmaxf(expf(scales[3 * i]), expf(scales[3 * i + 1])), expf(scales[3 * i + 2]));
    const bool clone = (ms < p.scale_thresh) && grad;                         // :152-153
    const bool split = (ms > p.scale_thresh) && grad;                         // :164-165
    const float sig = 1.0f / (1.0f + expf(-opacities[i]));
    const bool prune = ((sig < 0.1f) && (ms > 0.5f)) || split;                // :181-183
    flags[i] = (uint8_t)((clone ? TS_DENSIFY_CLONE : 0) | (split ? TS_DENSIFY_SPLIT : 0) |
                         (prune ? TS_DENSIFY_PRUNE : 0));
}

__device__ __forceinline__ packed_t packed_counts(uint8_t f) {
    return ((f & TS_DENSIFY_PRUNE) ? 0ll : 1ll) | ((f & TS_DENSIFY_CLONE) ? 1ll << kFieldBits : 0ll) |
           ((f & TS_DENSIFY_SPLIT) ? 1ll << (2 * kFieldBits) : 0ll);
}
__device__ __forceinline__ int field(packed_t v, int which) {
    return (int)((v >> (which * kFieldBits)) & kFieldMask);
}

__device__ __forceinline__ packed_t wave_inclusive_scan(packed_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const packed_t u = __shfl_up(v, d);
        if (lane >= d) v += u;
    }
    return v;
}

// inclusive scan of one packed word per thread over the 256-thread block; *total = block sum
__device__ __forceinline__ packed_t block_inclusive_scan(packed_t v, packed_t* total) {
    __shared__ packed_t wave_sum[kThreads / 64];
    const int wave = threadIdx.x >> 6;
    packed_t inc = wave_inclusive_scan(v);
    if ((threadIdx.x & 63) == 63) wave_sum[wave] = inc;
    __syncthreads();
    packed_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
        if (w < wave) base += wave_sum[w];
        all += wave_sum[w];
    }
    *total = all;
    __syncthreads();
    return inc + base;
}

__device__ __forceinline__ packed_t load_flags4(int n, const uint8_t* __restrict__ flags, int first,
                                                uint8_t* f) {
    packed_t sum = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        f[k] = (first + k < n) ? flags[first + k] : (uint8_t)TS_DENSIFY_PRUNE;   // out of range: nothing
        sum += (first + k < n) ? packed_counts(f[k]) : 0;
    }
    return sum;
}

// per block of 1024 Gaussians: (kept, cloned, split) counts -> ws[0..3 nb)
__global__ __launch_bounds__(kThreads) void count_kernel(int n, const uint8_t* __restrict__ flags,
                                                         int nb, int* __restrict__ ws) {
    uint8_t f[kItems];
    const int first = blockIdx.x * kBlockItems + threadIdx.x * kItems;
    const packed_t mine = load_flags4(n, flags, first, f);
    packed_t total;
    block_inclusive_scan(mine, &total);
    if (threadIdx.x == 0) {
        ws[blockIdx.x] = field(total, 0);
        ws[nb + blockIdx.x] = field(total, 1);
        ws[2 * nb + blockIdx.x] = field(total, 2);
    }
}

// single block: the three per-block count arrays -> exclusive bases in place; counts[4] = K, C, S, N'
__global__ __launch_bounds__(kThreads) void block_bases_kernel(int nb, int* __restrict__ ws,
                                                               int* __restrict__ counts) {
    __shared__ int carry;
    for (int a = 0; a < 3; ++a) {
        int* arr = ws + a * nb;
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (int b0 = 0; b0 < nb; b0 += kThreads) {
            const int b = b0 + threadIdx.x;
            const int v = b < nb ? arr[b] : 0;       // plain (unpacked) counts here: totals <= n < 2^31
            packed_t total;
            const int inc = (int)block_inclusive_scan((packed_t)v, &total);
            const int base = carry;
            if (b < nb) arr[b] = base + inc - v;
            __syncthreads();
            if (threadIdx.x == 0) carry = base + (int)total;
            __syncthreads();
        }
        if (threadIdx.x == 0) counts[a] = carry;
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[3] = counts[0] + counts[1] + 2 * counts[2];
}

// row map: src_of[kept | cloned | split, sample 0 | split, sample 1], source order inside each part
__global__ __launch_bounds__(kThreads) void plan_kernel(int n, const uint8_t* __restrict__ flags, int nb,
                                                        const int* __restrict__ ws,
                                                        const int* __restrict__ counts,
                                                        int* __restrict__ src_of) {
    uint8_t f[kItems];
    const int first = blockIdx.x * kBlockItems + threadIdx.x * kItems;
    const packed_t mine = load_flags4(n, flags, first, f);
    packed_t total;
    const packed_t exc = block_inclusive_scan(mine, &total) - mine;
    const int K = counts[0], C = counts[1], S = counts[2];
    int keep_at = ws[blockIdx.x] + field(exc, 0);
    int clone_at = K + ws[nb + blockIdx.x] + field(exc, 1);
    int split_at = K + C + ws[2 * nb + blockIdx.x] + field(exc, 2);
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int i = first + k;
        if (i >= n) break;
        if (!(f[k] & TS_DENSIFY_PRUNE)) src_of[keep_at++] = i;
        if (f[k] & TS_DENSIFY_CLONE) src_of[clone_at++] = i;
        if (f[k] & TS_DENSIFY_SPLIT) { src_of[split_at] = i; src_of[split_at + S] = i; ++split_at; }
    }
}

struct GatherTable {
    const float* src[TS_GATHER_MAX_TENSORS];
    float* dst[TS_GATHER_MAX_TENSORS];
    int width[TS_GATHER_MAX_TENSORS];       // floats per row
};

struct __attribute__((packed, aligned(4))) Unaligned4 { float x, y, z, w; };

// dst[r, :] = r < copy_rows ? src[src_of[r], :] : 0     (one thread per float, rows are 4..180 bytes)
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(GatherTable t, int dst_rows,
                                                               int copy_rows,
                                                               const int* __restrict__ src_of) {
    const int w = t.width[blockIdx.y];
    const float* __restrict__ src = t.src[blockIdx.y];
    float* __restrict__ dst = t.dst[blockIdx.y];
    const long long total = (long long)dst_rows * w;
    const long long stride = (long long)gridDim.x * kThreads;
    // element e -> (row, column): one double multiply and a +-1 fix-up instead of a 64-bit division per
    // float (the division was most of this kernel's instructions), then the columns of a unit are walked
    const double inv_w = 1.0 / (double)w;
    auto split = [&](long long e, int& r, int& c) {
        r = (int)((double)e * inv_w);
        long long cc = e - (long long)r * w;
        if (cc < 0) { --r; cc += w; } else if (cc >= w) { ++r; cc -= w; }
        c = (int)cc;
    };
    auto row_base = [&](int r) -> long long {
        return r < copy_rows ? (long long)src_of[r] * w : -1;
    };
    // four consecutive output floats per lane: one 16-byte store (the tensors are 16-byte aligned and
    // the unit starts at a multiple of four floats), four 4-byte gathers that mostly share a source row
    const long long units = total >> 2;
    for (long long u = (long long)blockIdx.x * kThreads + threadIdx.x; u < units; u += stride) {
        int r, c;
        split(u << 2, r, c);
        long long base = row_base(r);
        if (c + 4 <= w) {
            // the unit lies inside one source row (42 of 45 units of an SH row, every quaternion):
            // one 16-byte load at 4-byte alignment
            Unaligned4 q = {0.0f, 0.0f, 0.0f, 0.0f};
            if (base >= 0) q = *reinterpret_cast<const Unaligned4*>(src + base + c);
            reinterpret_cast<float4*>(dst)[u] = make_float4(q.x, q.y, q.z, q.w);
            continue;
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = base >= 0 ? src[base + c] : 0.0f;
            if (++c == w) { c = 0; ++r; if (k < 3) base = row_base(r); }
        }
        reinterpret_cast<float4*>(dst)[u] = make_float4(v[0], v[1], v[2], v[3]);
    }
    for (long long e = (units << 2) + (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += stride) {
        int r, c;
        split(e, r, c);
        const long long base = row_base(r);
        dst[e] = base >= 0 ? src[base + c] : 0.0f;
    }
}

// means / scales of the 2S sampled rows (GaussianDistribution.sample, :547-557; utils.py:41-73)
__global__ __launch_bounds__(kThreads) void split_fixup_kernel(int rows, const int* __restrict__ src_of,
                                                               const float* __restrict__ means,
                                                               const float* __restrict__ scales,
                                                               const float* __restrict__ quats,
                                                               const float* __restrict__ z,
                                                               float* __restrict__ means_out,
                                                               float* __restrict__ scales_out) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= rows) return;
    const int i = src_of[j];
    float e[3], pert[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        e[a] = expf(scales[3 * i + a]);
        pert[a] = z[3 * j + a] * e[a];                        // torch.normal(0, std) = z * std
        scales_out[3 * j + a] = logf(e[a] / 1.6f);            // :556
    }
    float qw = quats[4 * i], qx = quats[4 * i + 1], qy = quats[4 * i + 2], qz = quats[4 * i + 3];
    const float nrm = fmaxf(sqrtf(qw * qw + qx * qx + qy * qy + qz * qz), 1e-12f);   // F.normalize
    qw /= nrm; qx /= nrm; qy /= nrm; qz /= nrm;
    const float r00 = 1.0f - 2.0f * (qy * qy + qz * qz), r01 = 2.0f * (qx * qy - qw * qz),
                r02 = 2.0f * (qx * qz + qw * qy);
    const float r10 = 2.0f * (qx * qy + qw * qz), r11 = 1.0f - 2.0f * (qx * qx + qz * qz),
                r12 = 2.0f * (qy * qz - qw * qx);
    const float r20 = 2.0f * (qx * qz - qw * qy), r21 = 2.0f * (qy * qz + qw * qx),
                r22 = 1.0f - 2.0f * (qx * qx + qy * qy);
    means_out[3 * j] = (r00 * pert[0] + r01 * pert[1] + r02 * pert[2]) + means[3 * i];
    means_out[3 * j + 1] = (r10 * pert[0] + r11 * pert[1] + r12 * pert[2]) + means[3 * i + 1];
    means_out[3 * j + 2] = (r20 * pert[0] + r21 * pert[1] + r22 * pert[2]) + means[3 * i + 2];
}

}  // namespace

extern "C" {

int ts_grad_accum(int32_t n, const float* v_xy, float* accum, void* stream) {
    if (n < 0) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!v_xy || !accum) return TS_E_BADARG;
    hipLaunchKernelGGL(grad_accum_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, n, (const float2*)v_xy, accum);
    return launch_status();
}

int ts_densify_classify(int32_t n, const float* accum, const float* scales, const float* opacities,
                        const ts_densify_policy* policy, uint8_t* flags, void* stream) {
    if (n < 0 || !policy) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!accum || !scales || !opacities || !flags) return TS_E_BADARG;
    hipLaunchKernelGGL(classify_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, n, accum, scales, opacities, *policy, flags);
    return launch_status();
}

int64_t ts_densify_ws_ints(int32_t n) { return n <= 0 ? 4 : 3 * (int64_t)num_blocks(n) + 4; }

int ts_densify_plan(int32_t n, const uint8_t* flags, int32_t* ws, int32_t* counts, int32_t* src_of,
                    void* stream) {
    if (n < 0 || !ws || !counts) return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return (int)hipMemsetAsync(counts, 0, 4 * sizeof(int32_t), s);
    if (!flags || !src_of) return TS_E_BADARG;
    const int nb = num_blocks(n);
    hipLaunchKernelGGL(count_kernel, dim3(nb), dim3(kThreads), 0, s, n, flags, nb, ws);
    hipLaunchKernelGGL(block_bases_kernel, dim3(1), dim3(kThreads), 0, s, nb, ws, counts);
    hipLaunchKernelGGL(plan_kernel, dim3(nb), dim3(kThreads), 0, s, n, flags, nb, ws, counts, src_of);
    return launch_status();
}

int ts_gather_rows(int32_t num_tensors, const float* const* src, float* const* dst,
                   const int32_t* row_floats, int32_t dst_rows, int32_t copy_rows,
                   const int32_t* src_of, void* stream) {
    if (num_tensors < 0 || num_tensors > TS_GATHER_MAX_TENSORS || dst_rows < 0 || copy_rows < 0 ||
        copy_rows > dst_rows)
        return TS_E_BADARG;
    if (num_tensors == 0 || dst_rows == 0) return 0;
    if (!src || !dst || !row_floats || (copy_rows > 0 && !src_of)) return TS_E_BADARG;
    GatherTable t;
    long long most = 0;
    for (int i = 0; i < TS_GATHER_MAX_TENSORS; ++i) {
        const bool on = i < num_tensors;
        t.src[i] = on ? src[i] : nullptr; t.dst[i] = on ? dst[i] : nullptr;
        t.width[i] = on ? row_floats[i] : 0;
        if (!on) continue;
        if (row_floats[i] < 0) return TS_E_BADARG;
        if (row_floats[i] > 0 && (!dst[i] || (copy_rows > 0 && !src[i]))) return TS_E_BADARG;
        const long long e = (long long)dst_rows * row_floats[i];
        if (e > most) most = e;
    }
    if (most == 0) return 0;
    long long blocks = (most / 4 + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks, num_tensors), dim3(kThreads), 0,
                       (hipStream_t)stream, t, dst_rows, copy_rows, src_of);
    return launch_status();
}

int ts_split_fixup(int32_t rows, const int32_t* src_of_split, const float* means, const float* scales,
                   const float* quats, const float* z, float* means_out, float* scales_out,
                   void* stream) {
    if (rows < 0) return TS_E_BADARG;
    if (rows == 0) return 0;
    if (!src_of_split || !means || !scales || !quats || !z || !means_out || !scales_out)
        return TS_E_BADARG;
    hipLaunchKernelGGL(split_fixup_kernel, dim3((rows + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, rows, src_of_split, means, scales, quats, z, means_out,
                       scales_out);
    return launch_status();
}

}  // extern "C"
