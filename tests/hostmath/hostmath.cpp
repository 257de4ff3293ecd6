// Host-side checker of tinysplat_amd/csrc/splat_math.h.  TEST INFRASTRUCTURE: compiles the very
// same per-Gaussian math header the HIP kernels use with g++ (no FMA contraction), so the CPU test
// suite can compare it with the oracle without a GPU.  Never loaded by the product.
#include "../../tinysplat_amd/csrc/splat_math.h"

extern "C" {

struct hm_camera {
    float fx, fy, cx, cy;
    int W, H, tbx, tby, row0, rows;
    float gs, clip;
};

static ts::Cam make_cam(const float* viewmat, const float* projmat, const hm_camera* c) {
    ts::Cam C;
    for (int i = 0; i < 12; ++i) C.v[i] = viewmat[i];
    for (int i = 0; i < 16; ++i) C.p[i] = projmat[i];
    C.fx = c->fx; C.fy = c->fy; C.cx = c->cx; C.cy = c->cy;
    C.W = c->W; C.H = c->H; C.tbx = c->tbx; C.tby = c->tby; C.row0 = c->row0; C.rows = c->rows;
    C.gs = c->gs; C.clip = c->clip;
    return C;
}

void hm_project_fwd(int n, const float* means, const float* scales, const float* quats,
                    const float* viewmat, const float* projmat, const hm_camera* cam, float* xys,
                    float* depths, int* radii, float* conics, int* nth, float* cov3d) {
    const ts::Cam C = make_cam(viewmat, projmat, cam);
    for (int i = 0; i < n; ++i) {
        ts::ProjOut o;
        ts::project_one(C, means + 3 * i, scales + 3 * i, quats + 4 * i, o);
        xys[2 * i] = o.x; xys[2 * i + 1] = o.y; depths[i] = o.depth; radii[i] = o.radius;
        for (int k = 0; k < 3; ++k) conics[3 * i + k] = o.conic[k];
        nth[i] = o.tiles;
        for (int k = 0; k < 6; ++k) cov3d[6 * i + k] = o.cov3d[k];
    }
}

void hm_project_bwd(int n, const float* means, const float* scales, const float* quats,
                    const float* viewmat, const float* projmat, const hm_camera* cam,
                    const int* radii, const float* v_xy, const float* v_depth, const float* v_conic,
                    const float* v_cov3d, float* v_means, float* v_scales, float* v_quats) {
    const ts::Cam C = make_cam(viewmat, projmat, cam);
    for (int i = 0; i < n; ++i) {
        ts::ProjGrad g;
        for (int k = 0; k < 3; ++k) { g.v_mean[k] = 0; g.v_scale[k] = 0; }
        for (int k = 0; k < 4; ++k) g.v_quat[k] = 0;
        if (radii[i] > 0)
            ts::project_one_vjp(C, means + 3 * i, scales + 3 * i, quats + 4 * i, v_xy + 2 * i,
                                v_depth[i], v_conic + 3 * i, v_cov3d ? v_cov3d + 6 * i : nullptr, g);
        for (int k = 0; k < 3; ++k) { v_means[3 * i + k] = g.v_mean[k]; v_scales[3 * i + k] = g.v_scale[k]; }
        for (int k = 0; k < 4; ++k) v_quats[4 * i + k] = g.v_quat[k];
    }
}

void hm_sh_basis(int n, int degree, const float* dirs, float* Y) {
    const int nb = ts::sh_num_bases(degree);
    for (int i = 0; i < n; ++i)
        ts::sh_basis(degree, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], Y + (size_t)nb * i);
}

void hm_tile_bbox(int n, const float* xys, const float* radii, int tbx, int tby, int row0, int rows,
                  int* out4) {
    for (int i = 0; i < n; ++i) {
        const ts::TileBox b = ts::tile_bbox(xys[2 * i], xys[2 * i + 1], radii[i], tbx, tby, row0, rows);
        out4[4 * i] = b.minx; out4[4 * i + 1] = b.miny; out4[4 * i + 2] = b.maxx; out4[4 * i + 3] = b.maxy;
    }
}

}  // extern "C"
