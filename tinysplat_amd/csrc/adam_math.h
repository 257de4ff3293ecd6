// adam_math.h - the Adam update of ONE element (torch.optim.Adam defaults: no amsgrad, no weight decay;
// /root/reference/scripts/train.py:26, model_gaussian.py:112-120), shared by the stand-alone optimiser launch
// (train.hip: ts_adam_step) and by the parameter-stage backward kernels that apply it to the gradient they hold in
// registers (project.hip: ts_sh_colors_bwd_adam, ts_project_bwd_adam).  Contraction is pinned, so that every caller -
// whatever its translation unit's flags - rounds alike: the fused step is bit for bit the two-launch step.
#pragma once
#include <math.h>

namespace ts {

struct AdamCoef {
    float beta1, beta2, eps;
    float step_size;     // lr / (1 - beta1^step)
    float bc2_sqrt;      // sqrt(1 - beta2^step)
};

// double-precision bias corrections as torch.optim.Adam computes them on the host; `step` is the tensor's own 1-based
// step count (torch keeps it per parameter)
inline AdamCoef adam_coef(float lr, int step, float beta1, float beta2, float eps) {
    AdamCoef c;
    c.beta1 = beta1; c.beta2 = beta2; c.eps = eps;
    const double b1p = __builtin_pow((double)beta1, (double)step);
    const double b2p = __builtin_pow((double)beta2, (double)step);
    c.step_size = (float)((double)lr / (1.0 - b1p));
    c.bc2_sqrt = (float)__builtin_sqrt(1.0 - b2p);
    return c;
}

#if defined(__HIPCC__)
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const AdamCoef& c) {
#pragma clang fp contract(off)
    const float mn = c.beta1 * m + (1.0f - c.beta1) * g;
    const float vn = c.beta2 * v + ((1.0f - c.beta2) * g) * g;
    m = mn;
    v = vn;
    p = p - c.step_size * (mn / (sqrtf(vn) / c.bc2_sqrt + c.eps));
}
#endif

}  // namespace ts
