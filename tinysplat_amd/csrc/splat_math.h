// splat_math.h - per-Gaussian arithmetic shared by the HIP kernels (device) and by the host-side
// math checker built from tests/ (g++).  Everything here is scalar, branch-light code that a kernel
// calls with one Gaussian per lane.
//
// The projection follows the gsplat 0.1.x semantics that tinysplat's call site relies on
// (/root/reference/tinysplat/splatting/rasterize.py:32, args rasterize.py:64-73) with the exact
// association order of oracle/gsplat_oracle.py::project_gaussians, so that a translation unit
// compiled with -ffp-contract=off reproduces the float32 oracle bit-for-bit.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TS_HD __host__ __device__ __forceinline__
#else
#define TS_HD inline
#endif

namespace ts {

constexpr float kBlur = 0.3f;          // low-pass added to the cov2d diagonal
// SURVEY App. C's open choices in the device code are compile-time switches, so that vectors from a pinned gsplat are
// a one-line flip; the other setting of each is built and checked against the oracle with the same constants once per
// test session (tests/test_gpu_variants.py).  (App. C #9, the SH view directions, is a run-time flag of the adapter:
// GaussianRasterizer(correct_viewdirs=True).)
//   TS_PIX_OFF (0 | 0.5f)          pixel (j, i) sampled at (j + off, i + off); 0.1.3-era: 0
//   TS_BWD_CLAMP_UPSTREAM (0 | 1)  0: the backward pass differentiates the forward pass (alpha clamped at 0.999, no
//                                  gradient through a clamped alpha); 1: upstream's rasterize_backward as SURVEY App. C
//                                  records it - alpha re-clamped at 0.99, v_sigma = -opacity * vis * v_alpha unconditionally
#ifndef TS_PIX_OFF
#define TS_PIX_OFF 0.0f
#endif
#ifndef TS_BWD_CLAMP_UPSTREAM
#define TS_BWD_CLAMP_UPSTREAM 0
#endif
//   TS_FOV_CLAMP_BWD_UNGATED (0 | 1)  App. C #4.  0: the projection's backward pass differentiates the forward pass - a
//                                  view-space coordinate the 1.3 tan(fov) clamp caught passes no gradient (its value
//                                  lim * z sends it to z instead); 1: upstream's project_cov3d_ewa_vjp as SURVEY App. A.6
//                                  recalls it - J is evaluated at the clamped t and v_t goes to the view-space mean as if
//                                  no clamp had acted (a straight-through clamp)
#ifndef TS_FOV_CLAMP_BWD_UNGATED
#define TS_FOV_CLAMP_BWD_UNGATED 0
#endif
constexpr float kAlphaMax = 0.999f;    // forward (and, by default, backward) alpha clamp
constexpr float kAlphaMaxBwd = TS_BWD_CLAMP_UPSTREAM ? 0.99f : kAlphaMax;
constexpr bool kClampGatesGrad = !TS_BWD_CLAMP_UPSTREAM;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTEps = 1e-4f;         // transmittance early-out
constexpr float kPixOff = TS_PIX_OFF;
constexpr int kTile = 16;

struct Cam {                 // device-side camera block (matrices by value)
    float v[12];             // 3x4 view
    float p[16];             // 4x4 proj @ view
    float fx, fy, cx, cy;
    int W, H, tbx, tby, row0, rows;
    float gs, clip;
};

struct TileBox { int minx, miny, maxx, maxy; };

// Tile rectangle [min,max) of (centre, radius): trunc-toward-zero then clamp, as upstream's
// get_tile_bbox; rows additionally clipped to the stripe [row0, row0+rows).
TS_HD TileBox tile_bbox(float x, float y, float radius, int tbx, int tby, int row0, int rows) {
    const float tcx = x / 16.0f, tcy = y / 16.0f, tr = radius / 16.0f;
    TileBox b;
    b.minx = (int)fminf(fmaxf(truncf(tcx - tr), 0.0f), (float)tbx);
    b.maxx = (int)fminf(fmaxf(truncf(tcx + tr + 1.0f), 0.0f), (float)tbx);
    b.miny = (int)fminf(fmaxf(truncf(tcy - tr), 0.0f), (float)tby);
    b.maxy = (int)fminf(fmaxf(truncf(tcy + tr + 1.0f), 0.0f), (float)tby);
    b.miny = b.miny > row0 ? b.miny : row0;
    const int r1 = row0 + rows;
    b.maxy = b.maxy < r1 ? b.maxy : r1;
    return b;
}

struct Rot { float r00, r01, r02, r10, r11, r12, r20, r21, r22; };

// (w,x,y,z) -> R after normalisation; formula of tinysplat/utils.py:41-73.
TS_HD Rot quat_to_rot(float qw, float qx, float qy, float qz, float* inv_norm_out = nullptr,
                      float* nq = nullptr) {
    const float n = sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
    const float w = qw / n, x = qx / n, y = qy / n, z = qz / n;
    if (inv_norm_out) *inv_norm_out = 1.0f / n;
    if (nq) { nq[0] = w; nq[1] = x; nq[2] = y; nq[3] = z; }
    Rot r;
    r.r00 = 1.0f - 2.0f * (y * y + z * z);
    r.r01 = 2.0f * (x * y - w * z);
    r.r02 = 2.0f * (x * z + w * y);
    r.r10 = 2.0f * (x * y + w * z);
    r.r11 = 1.0f - 2.0f * (x * x + z * z);
    r.r12 = 2.0f * (y * z - w * x);
    r.r20 = 2.0f * (x * z - w * y);
    r.r21 = 2.0f * (y * z + w * x);
    r.r22 = 1.0f - 2.0f * (x * x + y * y);
    return r;
}

struct ProjOut {
    float x, y, depth, conic[3], cov3d[6];
    int radius, tiles;
};

// Intermediates shared by forward and backward.
struct ProjMid {
    float px, py, pz;              // view-space mean
    float tx, ty;                  // fov-clamped view x,y
    bool clamp_x, clamp_y;
    float sgn_x, sgn_y;            // value of the clamp (+-lim) when clamped
    float rz, rz2;
    float j00, j02, j11, j12;
    float t00, t01, t02, t10, t11, t12;   // T = J W
    float c00, c01, c02, c11, c12, c22;   // cov3d
    float a, b, c, det;                   // cov2d (+blur), determinant
    float m00, m01, m02, m10, m11, m12, m20, m21, m22;  // M = R S
    Rot R;
    float s0, s1, s2;              // glob_scale * scale
};

TS_HD bool project_mid(const Cam& C, const float m[3], const float sc[3], const float q[4],
                       ProjMid& o) {
    const float* V = C.v;
    o.px = ((V[0] * m[0] + V[1] * m[1]) + V[2] * m[2]) + V[3];
    o.py = ((V[4] * m[0] + V[5] * m[1]) + V[6] * m[2]) + V[7];
    o.pz = ((V[8] * m[0] + V[9] * m[1]) + V[10] * m[2]) + V[11];
    if (!(o.pz > C.clip)) return false;

    o.R = quat_to_rot(q[0], q[1], q[2], q[3]);
    o.s0 = C.gs * sc[0]; o.s1 = C.gs * sc[1]; o.s2 = C.gs * sc[2];
    o.m00 = o.R.r00 * o.s0; o.m01 = o.R.r01 * o.s1; o.m02 = o.R.r02 * o.s2;
    o.m10 = o.R.r10 * o.s0; o.m11 = o.R.r11 * o.s1; o.m12 = o.R.r12 * o.s2;
    o.m20 = o.R.r20 * o.s0; o.m21 = o.R.r21 * o.s1; o.m22 = o.R.r22 * o.s2;
    o.c00 = (o.m00 * o.m00 + o.m01 * o.m01) + o.m02 * o.m02;
    o.c01 = (o.m00 * o.m10 + o.m01 * o.m11) + o.m02 * o.m12;
    o.c02 = (o.m00 * o.m20 + o.m01 * o.m21) + o.m02 * o.m22;
    o.c11 = (o.m10 * o.m10 + o.m11 * o.m11) + o.m12 * o.m12;
    o.c12 = (o.m10 * o.m20 + o.m11 * o.m21) + o.m12 * o.m22;
    o.c22 = (o.m20 * o.m20 + o.m21 * o.m21) + o.m22 * o.m22;

    const float limx = 1.3f * ((0.5f * (float)C.W) / C.fx);
    const float limy = 1.3f * ((0.5f * (float)C.H) / C.fy);
    const float qx = o.px / o.pz, qy = o.py / o.pz;
    const float cqx = fminf(limx, fmaxf(-limx, qx));
    const float cqy = fminf(limy, fmaxf(-limy, qy));
    o.clamp_x = (cqx != qx); o.clamp_y = (cqy != qy);
    o.sgn_x = cqx; o.sgn_y = cqy;
    o.tx = o.pz * cqx;
    o.ty = o.pz * cqy;
    o.rz = 1.0f / o.pz;
    o.rz2 = o.rz * o.rz;
    o.j00 = C.fx * o.rz;
    o.j02 = -(C.fx * o.tx) * o.rz2;
    o.j11 = C.fy * o.rz;
    o.j12 = -(C.fy * o.ty) * o.rz2;
    o.t00 = o.j00 * V[0] + o.j02 * V[8];
    o.t01 = o.j00 * V[1] + o.j02 * V[9];
    o.t02 = o.j00 * V[2] + o.j02 * V[10];
    o.t10 = o.j11 * V[4] + o.j12 * V[8];
    o.t11 = o.j11 * V[5] + o.j12 * V[9];
    o.t12 = o.j11 * V[6] + o.j12 * V[10];
    const float u0 = (o.t00 * o.c00 + o.t01 * o.c01) + o.t02 * o.c02;
    const float u1 = (o.t00 * o.c01 + o.t01 * o.c11) + o.t02 * o.c12;
    const float u2 = (o.t00 * o.c02 + o.t01 * o.c12) + o.t02 * o.c22;
    const float w0 = (o.t10 * o.c00 + o.t11 * o.c01) + o.t12 * o.c02;
    const float w1 = (o.t10 * o.c01 + o.t11 * o.c11) + o.t12 * o.c12;
    const float w2 = (o.t10 * o.c02 + o.t11 * o.c12) + o.t12 * o.c22;
    o.a = ((u0 * o.t00 + u1 * o.t01) + u2 * o.t02) + kBlur;
    o.b = (u0 * o.t10 + u1 * o.t11) + u2 * o.t12;
    o.c = ((w0 * o.t10 + w1 * o.t11) + w2 * o.t12) + kBlur;
    o.det = o.a * o.c - o.b * o.b;
    return true;
}

// Full forward for one Gaussian.  Returns false (and zeroed outputs) when culled.
TS_HD bool project_one(const Cam& C, const float m[3], const float sc[3], const float q[4],
                       ProjOut& out) {
    out.x = out.y = out.depth = 0.0f;
    out.conic[0] = out.conic[1] = out.conic[2] = 0.0f;
    for (int i = 0; i < 6; ++i) out.cov3d[i] = 0.0f;
    out.radius = 0; out.tiles = 0;
    ProjMid o;
    if (!project_mid(C, m, sc, q, o)) return false;
    out.cov3d[0] = o.c00; out.cov3d[1] = o.c01; out.cov3d[2] = o.c02;
    out.cov3d[3] = o.c11; out.cov3d[4] = o.c12; out.cov3d[5] = o.c22;
    if (o.det == 0.0f) return false;
    const float inv_det = 1.0f / o.det;
    const float mid = 0.5f * (o.a + o.c);
    const float sq = sqrtf(fmaxf(0.1f, mid * mid - o.det));
    const float lam = fmaxf(mid + sq, mid - sq);
    const float radius = ceilf(3.0f * sqrtf(lam));

    const float* P = C.p;
    const float hx = ((P[0] * m[0] + P[1] * m[1]) + P[2] * m[2]) + P[3];
    const float hy = ((P[4] * m[0] + P[5] * m[1]) + P[6] * m[2]) + P[7];
    const float hw = ((P[12] * m[0] + P[13] * m[1]) + P[14] * m[2]) + P[15];
    const float rw = 1.0f / (hw + 1e-6f);
    const float x = ((0.5f * (float)C.W) * (hx * rw) + C.cx) - 0.5f;
    const float y = ((0.5f * (float)C.H) * (hy * rw) + C.cy) - 0.5f;

    // "no tile hit" is decided on the full frame so that radii / xys do not depend on the stripe;
    // num_tiles_hit counts the stripe's tiles only.
    const TileBox full = tile_bbox(x, y, radius, C.tbx, C.tby, 0, C.tby);
    if ((full.maxx - full.minx) * (full.maxy - full.miny) <= 0) return false;
    const TileBox b = tile_bbox(x, y, radius, C.tbx, C.tby, C.row0, C.rows);
    const int h = b.maxy - b.miny;
    out.tiles = h > 0 ? (b.maxx - b.minx) * h : 0;
    out.x = x; out.y = y; out.depth = o.pz;
    out.conic[0] = o.c * inv_det; out.conic[1] = -o.b * inv_det; out.conic[2] = o.a * inv_det;
    out.radius = (int)radius;
    return true;
}

struct ProjGrad { float v_mean[3], v_scale[3], v_quat[4]; };

// VJP of project_one.  v_conic: true partials w.r.t. the three stored conic entries; v_cov3d (6,
// may be null): partials w.r.t. the six stored cov3d entries.  Mathematically this is the
// gradient autograd derives from the oracle's forward: the fov clamp passes no gradient through a
// clamped coordinate, and the internal quaternion normalisation is differentiated.
TS_HD void project_one_vjp(const Cam& C, const float m[3], const float sc[3], const float q[4],
                           const float v_xy[2], float v_depth, const float v_conic[3],
                           const float* v_cov3d, ProjGrad& g) {
    for (int i = 0; i < 3; ++i) { g.v_mean[i] = 0.0f; g.v_scale[i] = 0.0f; }
    for (int i = 0; i < 4; ++i) g.v_quat[i] = 0.0f;
    ProjMid o;
    if (!project_mid(C, m, sc, q, o)) return;
    if (o.det == 0.0f) return;
    const float* V = C.v;
    const float* P = C.p;

    // pixel centre
    const float hx = ((P[0] * m[0] + P[1] * m[1]) + P[2] * m[2]) + P[3];
    const float hy = ((P[4] * m[0] + P[5] * m[1]) + P[6] * m[2]) + P[7];
    const float hw = ((P[12] * m[0] + P[13] * m[1]) + P[14] * m[2]) + P[15];
    const float rw = 1.0f / (hw + 1e-6f);
    const float vnx = 0.5f * (float)C.W * v_xy[0], vny = 0.5f * (float)C.H * v_xy[1];
    const float v_hx = vnx * rw, v_hy = vny * rw;
    const float v_hw = -(vnx * hx + vny * hy) * rw * rw;
    float vm0 = P[0] * v_hx + P[4] * v_hy + P[12] * v_hw;
    float vm1 = P[1] * v_hx + P[5] * v_hy + P[13] * v_hw;
    float vm2 = P[2] * v_hx + P[6] * v_hy + P[14] * v_hw;

    // conic -> cov2d : v_Sigma = -X G X with X the conic matrix, G the symmetric gradient matrix
    const float inv_det = 1.0f / o.det;
    const float X00 = o.c * inv_det, X01 = -o.b * inv_det, X11 = o.a * inv_det;
    const float G00 = v_conic[0], G01 = 0.5f * v_conic[1], G11 = v_conic[2];
    const float Y00 = X00 * G00 + X01 * G01, Y01 = X00 * G01 + X01 * G11;   // X G
    const float Y10 = X01 * G00 + X11 * G01, Y11 = X01 * G01 + X11 * G11;
    const float S00 = -(Y00 * X00 + Y01 * X01);                             // -(X G X), symmetric
    const float S01 = -(Y00 * X01 + Y01 * X11);
    const float S11 = -(Y10 * X01 + Y11 * X11);

    // cov2d = T C T^T : Vc = T^T S T (3x3 symmetric gradient matrix of cov3d), v_T = 2 S T C
    const float a0 = S00 * o.t00 + S01 * o.t10, a1 = S00 * o.t01 + S01 * o.t11,
                a2 = S00 * o.t02 + S01 * o.t12;                             // (S T) row 0
    const float b0 = S01 * o.t00 + S11 * o.t10, b1 = S01 * o.t01 + S11 * o.t11,
                b2 = S01 * o.t02 + S11 * o.t12;                             // (S T) row 1
    float vc00 = o.t00 * a0 + o.t10 * b0, vc01 = o.t00 * a1 + o.t10 * b1,
          vc02 = o.t00 * a2 + o.t10 * b2, vc11 = o.t01 * a1 + o.t11 * b1,
          vc12 = o.t01 * a2 + o.t11 * b2, vc22 = o.t02 * a2 + o.t12 * b2;
    if (v_cov3d) {
        vc00 += v_cov3d[0]; vc01 += 0.5f * v_cov3d[1]; vc02 += 0.5f * v_cov3d[2];
        vc11 += v_cov3d[3]; vc12 += 0.5f * v_cov3d[4]; vc22 += v_cov3d[5];
    }
    // v_T = 2 (S T) C
    const float vt00 = 2.0f * (a0 * o.c00 + a1 * o.c01 + a2 * o.c02);
    const float vt01 = 2.0f * (a0 * o.c01 + a1 * o.c11 + a2 * o.c12);
    const float vt02 = 2.0f * (a0 * o.c02 + a1 * o.c12 + a2 * o.c22);
    const float vt10 = 2.0f * (b0 * o.c00 + b1 * o.c01 + b2 * o.c02);
    const float vt11 = 2.0f * (b0 * o.c01 + b1 * o.c11 + b2 * o.c12);
    const float vt12 = 2.0f * (b0 * o.c02 + b1 * o.c12 + b2 * o.c22);
    // T = J W  ->  v_J = v_T W^T
    const float vj00 = vt00 * V[0] + vt01 * V[1] + vt02 * V[2];
    const float vj02 = vt00 * V[8] + vt01 * V[9] + vt02 * V[10];
    const float vj11 = vt10 * V[4] + vt11 * V[5] + vt12 * V[6];
    const float vj12 = vt10 * V[8] + vt11 * V[9] + vt12 * V[10];
    // J(tx, ty, pz)
    const float rz3 = o.rz2 * o.rz;
    const float v_tx = -C.fx * o.rz2 * vj02;
    const float v_ty = -C.fy * o.rz2 * vj12;
    float v_pz = -C.fx * o.rz2 * vj00 - C.fy * o.rz2 * vj11
                 + 2.0f * C.fx * o.tx * rz3 * vj02 + 2.0f * C.fy * o.ty * rz3 * vj12;
    float v_px = 0.0f, v_py = 0.0f;
    if (TS_FOV_CLAMP_BWD_UNGATED) {
        v_px = v_tx; v_py = v_ty;               // App. C #4, upstream's reading: the clamp is invisible to the VJP
    } else {
        if (o.clamp_x) v_pz += o.sgn_x * v_tx; else v_px = v_tx;
        if (o.clamp_y) v_pz += o.sgn_y * v_ty; else v_py = v_ty;
    }
    v_pz += v_depth;
    vm0 += V[0] * v_px + V[4] * v_py + V[8] * v_pz;
    vm1 += V[1] * v_px + V[5] * v_py + V[9] * v_pz;
    vm2 += V[2] * v_px + V[6] * v_py + V[10] * v_pz;
    g.v_mean[0] = vm0; g.v_mean[1] = vm1; g.v_mean[2] = vm2;

    // cov3d = M M^T : v_M = 2 Vc M
    const float vM00 = 2.0f * (vc00 * o.m00 + vc01 * o.m10 + vc02 * o.m20);
    const float vM01 = 2.0f * (vc00 * o.m01 + vc01 * o.m11 + vc02 * o.m21);
    const float vM02 = 2.0f * (vc00 * o.m02 + vc01 * o.m12 + vc02 * o.m22);
    const float vM10 = 2.0f * (vc01 * o.m00 + vc11 * o.m10 + vc12 * o.m20);
    const float vM11 = 2.0f * (vc01 * o.m01 + vc11 * o.m11 + vc12 * o.m21);
    const float vM12 = 2.0f * (vc01 * o.m02 + vc11 * o.m12 + vc12 * o.m22);
    const float vM20 = 2.0f * (vc02 * o.m00 + vc12 * o.m10 + vc22 * o.m20);
    const float vM21 = 2.0f * (vc02 * o.m01 + vc12 * o.m11 + vc22 * o.m21);
    const float vM22 = 2.0f * (vc02 * o.m02 + vc12 * o.m12 + vc22 * o.m22);
    // M = R diag(gs*scale)
    g.v_scale[0] = C.gs * (o.R.r00 * vM00 + o.R.r10 * vM10 + o.R.r20 * vM20);
    g.v_scale[1] = C.gs * (o.R.r01 * vM01 + o.R.r11 * vM11 + o.R.r21 * vM21);
    g.v_scale[2] = C.gs * (o.R.r02 * vM02 + o.R.r12 * vM12 + o.R.r22 * vM22);
    const float vR00 = vM00 * o.s0, vR01 = vM01 * o.s1, vR02 = vM02 * o.s2;
    const float vR10 = vM10 * o.s0, vR11 = vM11 * o.s1, vR12 = vM12 * o.s2;
    const float vR20 = vM20 * o.s0, vR21 = vM21 * o.s1, vR22 = vM22 * o.s2;
    // R(q_hat)
    float inv_n, nq[4];
    quat_to_rot(q[0], q[1], q[2], q[3], &inv_n, nq);
    const float w = nq[0], x = nq[1], y = nq[2], z = nq[3];
    const float gw = 2.0f * (-z * vR01 + y * vR02 + z * vR10 - x * vR12 - y * vR20 + x * vR21);
    const float gx = 2.0f * (y * vR01 + z * vR02 + y * vR10 - 2.0f * x * vR11 - w * vR12
                             + z * vR20 + w * vR21 - 2.0f * x * vR22);
    const float gy = 2.0f * (-2.0f * y * vR00 + x * vR01 + w * vR02 + x * vR10 + z * vR12
                             - w * vR20 + z * vR21 - 2.0f * y * vR22);
    const float gz = 2.0f * (-2.0f * z * vR00 - w * vR01 + x * vR02 + w * vR10 - 2.0f * z * vR11
                             + y * vR12 + x * vR20 + y * vR21);
    // q_hat = q / |q|
    const float dotp = w * gw + x * gx + y * gy + z * gz;
    g.v_quat[0] = (gw - w * dotp) * inv_n;
    g.v_quat[1] = (gx - x * dotp) * inv_n;
    g.v_quat[2] = (gy - y * dotp) * inv_n;
    g.v_quat[3] = (gz - z * dotp) * inv_n;
}

// ------------------------------------------------------------------------------------------------
// Spherical harmonics (gsplat.sh.spherical_harmonics, call site rasterize.py:38).
// Writes the basis values for bands <= degree into Y[0 .. (degree+1)^2).
// ------------------------------------------------------------------------------------------------
TS_HD int sh_num_bases(int degree) {
    return degree == 0 ? 1 : degree == 1 ? 4 : degree == 2 ? 9 : degree == 3 ? 16 : 25;
}

TS_HD void sh_basis(int degree, float dx, float dy, float dz, float* Y) {
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / n, y = dy / n, z = dz / n;
    Y[0] = 0.28209479177387814f;
    if (degree < 1) return;
    Y[1] = -0.4886025119029199f * y;
    Y[2] = 0.4886025119029199f * z;
    Y[3] = -0.4886025119029199f * x;
    if (degree < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    Y[4] = 1.0925484305920792f * xy;
    Y[5] = -1.0925484305920792f * yz;
    Y[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    Y[7] = -1.0925484305920792f * xz;
    Y[8] = 0.5462742152960396f * (xx - yy);
    if (degree < 3) return;
    Y[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
    Y[10] = 2.890611442640554f * xy * z;
    Y[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
    Y[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    Y[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
    Y[14] = 1.445305721320277f * z * (xx - yy);
    Y[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
    if (degree < 4) return;
    Y[16] = 2.5033429417967046f * xy * (xx - yy);
    Y[17] = -1.7701307697799304f * yz * (3.0f * xx - yy);
    Y[18] = 0.9461746957575601f * xy * (7.0f * zz - 1.0f);
    Y[19] = -0.6690465435572892f * yz * (7.0f * zz - 3.0f);
    Y[20] = 0.10578554691520431f * (zz * (35.0f * zz - 30.0f) + 3.0f);
    Y[21] = -0.6690465435572892f * xz * (7.0f * zz - 3.0f);
    Y[22] = 0.47308734787878004f * (xx - yy) * (7.0f * zz - 1.0f);
    Y[23] = -1.7701307697799304f * xz * (xx - 3.0f * yy);
    Y[24] = 0.6258357354491761f * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
}

// ---- conservative "can this Gaussian reach that pixel rectangle" test ---------------------------
// Used by the compositing kernels (per 8x8 block, raster.hip: stage_splat) and by the tight tile
// binning (per 16x16 tile, binning.hip).  Works in the log2 domain: with hA = log2e/2 * conic.x,
// B = log2e * conic.y, hC = log2e/2 * conic.z the exponent is sigma' = hA dx^2 + B dx dy + hC dy^2 and
// alpha >= 1/255  <=>  sigma' <= log2(opacity) + log2(255).
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLog2_255 = 7.994353436858858f;

// Minimum over the rectangle dx in [xlo,xhi], dy in [ylo,yhi] of hA dx^2 + B dx dy + hC dy^2
// (hA, hC > 0 assumed by the caller; inv2A = 0.5 / hA, inv2C = 0.5 / hC).
TS_HD float min_form_on_rect(float hA, float B, float hC, float inv2A, float inv2C, float xlo,
                             float xhi, float ylo, float yhi) {
    if (xlo <= 0.0f && xhi >= 0.0f && ylo <= 0.0f && yhi >= 0.0f) return 0.0f;
    float best = 3.0e38f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float dx = e ? xhi : xlo;
        const float dy = fminf(fmaxf(-B * dx * inv2C, ylo), yhi);
        best = fminf(best, hA * dx * dx + dy * (B * dx + hC * dy));
        const float ey = e ? yhi : ylo;
        const float ex = fminf(fmaxf(-B * ey * inv2A, xlo), xhi);
        best = fminf(best, hC * ey * ey + ex * (B * ey + hA * ex));
    }
    return best;
}

// True unless NO sample position of the rectangle [x0,x1] x [y0,y1] can have alpha >= 1/255 for the
// Gaussian centred at (gx, gy).  tau = log2(opacity) + log2(255).  The slack (0.02 in the exponent,
// i.e. 1.4 % in alpha, plus a relative term for large offsets) covers the rounding of this bound and
// of the per-pixel evaluation, so a false return is a proof that every pixel's alpha test fails.
TS_HD bool rect_may_contribute(float hA, float B, float hC, float inv2A, float inv2C, float tau,
                               float gx, float gy, float x0, float x1, float y0, float y1) {
    const float xlo = gx - x1, xhi = gx - x0;
    const float ylo = gy - y1, yhi = gy - y0;
    const float m = min_form_on_rect(hA, B, hC, inv2A, inv2C, xlo, xhi, ylo, yhi);
    const float dxm = fmaxf(fabsf(xlo), fabsf(xhi));
    const float dym = fmaxf(fabsf(ylo), fabsf(yhi));
    const float mag = hA * dxm * dxm + hC * dym * dym + fabsf(B) * dxm * dym;
    return m <= tau + 0.02f + 4.0e-6f * mag;
}

}  // namespace ts
