// frame.hip - whole-frame executor: the render adapter's recipe
// (/root/reference/tinysplat/splatting/rasterize.py:26-62: project -> SH colours -> rasterize RGB
// [+ depth], and its backward) enqueued from native code.
//
// Nothing is computed here: every function below calls the per-stage C-ABI entries of this library
// (project.hip, binning.hip, raster.hip) in the order frame.py used to call them one by one through
// ctypes.  What it buys is host time: a frame is ~30 kernel launches, and issuing them from Python
// costs ~0.45 ms per frame - more than the GPU needs for a 100 k-Gaussian scene or for one tile
// stripe of a multi-GPU frame.  From here a launch costs ~2 us.  The one host round trip of the path
// (the intersection count that sizes the per-intersection buffers) stays with the caller: the scan kernel
// stores it into the caller's mapped pinned word (ts_frame_fwd_project), where the caller polls it between
// _prepare and _composite.
#include <hip/hip_runtime.h>

#include <cstdlib>
// roctx is optional: a ROCm install without rocprofiler-sdk still builds and loads the library (ranges become no-ops)
#if !defined(TS_NO_ROCTX) && __has_include(<rocprofiler-sdk-roctx/roctx.h>)
#include <rocprofiler-sdk-roctx/roctx.h>
#define TS_HAVE_ROCTX 1
#else
#define TS_HAVE_ROCTX 0
#endif

#include "../../include/tinysplat_hip.h"

// roctx ranges around the five executor calls: `rocprofv3 --marker-trace --kernel-trace` shows which kernels a
// call enqueued and the gaps between them (no cost without a tool attached)
struct TsRange {
#if TS_HAVE_ROCTX
    explicit TsRange(const char* name) { roctxRangePush(name); }
    ~TsRange() { roctxRangePop(); }
#else
    explicit TsRange(const char*) {}
#endif
};

#define TS_TRY(call)                 \
    do {                             \
        const int e_ = (call);       \
        if (e_ != 0) return e_;      \
    } while (0)

namespace {
inline bool bad(const ts_frame* f) {
    return !f || f->n < 0 || (f->channels != 3 && f->channels != 4) || f->num_bases < 1;
}
inline int num_tiles(const ts_frame* f) { return ts_num_tiles(&f->cam); }
// device address of a pinned host word, or null.  The last SUCCESSFUL answer is kept per (device, host address):
// a host thread uses one word per device; a failed lookup is not remembered.
inline int32_t* mapped_pointer(int32_t* host) {
    static thread_local int32_t* last_host = nullptr;
    static thread_local int32_t* last_dev = nullptr;
    static thread_local int last_device = -1;
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    if (host == last_host && device == last_device) return last_dev;
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess || !dev) {
        (void)hipGetLastError();
        return nullptr;
    }
    last_host = host;
    last_device = device;
    last_dev = static_cast<int32_t*>(dev);
    return last_dev;
}
// list segments (ts_camera.hints bits 8..11) replace the split blocks in the BACKWARD pass of a small launch
inline bool segmented(const ts_frame* f) {
    return ((f->cam.hints >> 8) & 15) > 1 && !f->cam.wide_tiles && !(f->flags & TS_FRAME_NARROW_WAVES);
}
inline int bwd_split(const ts_frame* f) {
    return ((f->flags & TS_FRAME_SPLIT) && !segmented(f)) ? TS_RASTER_SPLIT_BLOCKS : 0;
}
// Gaussians up to which a rank's owner stage runs its small-N fusions (TS_SMALL_N_FUSED, read once; 0 = never)
inline int small_n_fused() {
    static const int v = [] {
        const char* e = getenv("TS_SMALL_N_FUSED");
        return e ? atoi(e) : 262144;
    }();
    return v;
}
inline int raster_flags(const ts_frame* f) {
    return TS_RASTER_CLAMP_RGB | ((f->flags & TS_FRAME_SPLIT) ? TS_RASTER_SPLIT_BLOCKS : 0) |
           ((f->flags & TS_FRAME_NARROW_WAVES) ? TS_RASTER_NARROW_WAVES : 0);
}
}  // namespace

extern "C" {

int32_t ts_frame_struct_bytes(void) { return (int32_t)sizeof(ts_frame); }

int ts_frame_fwd_project(const ts_frame* f, void* stream) {
    TsRange range_("ts_frame_fwd_project");
    if (bad(f)) return TS_E_BADARG;
    // flags 3: log-scales and raw quaternions go in as they are (rasterize.py:72-73 folded into the kernel);
    // cov3d is not produced (the adapter discards it, rasterize.py:32)
    TS_TRY(ts_project_fwd(f->n, f->means, f->scales, f->quats, f->view34, f->projview, &f->cam, 3, f->xys,
                          f->depths, f->radii, f->conics, f->num_tiles_hit, nullptr, stream));
    // the count goes to the caller's pinned word from the scan kernel itself when that memory is mapped
    // into the device's address space (hipHostMalloc'd memory is); otherwise by a 4-byte copy
    int32_t* total_dev = f->total_host ? mapped_pointer(f->total_host) : nullptr;
    TS_TRY(ts_scan_tiles(f->n, f->num_tiles_hit, f->cum_tiles_hit, f->scan_ws, total_dev, stream));
    if (f->n > 0 && f->total_host && !total_dev) {
        const hipError_t e = hipMemcpyAsync(f->total_host, f->cum_tiles_hit + (f->n - 1), sizeof(int32_t),
                                            hipMemcpyDeviceToHost, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

int ts_frame_fwd_prepare(const ts_frame* f, void* stream) {
    TsRange range_("ts_frame_fwd_prepare");
    if (bad(f)) return TS_E_BADARG;
    // colour stage and record packing in one launch; channel 3 of an RGB + depth frame is the depth itself
    // (rasterize.py:48-50), taken from `depths`
    TS_TRY(ts_colors_pack_fwd(f->n, f->sh_degree, f->num_bases, f->means, f->origin, f->colors_dc,
                              f->num_bases > 1 ? f->colors_rest : nullptr, f->sh_mask,
                              (f->flags & TS_FRAME_STRIPE) ? f->num_tiles_hit : nullptr, f->channels,
                              TS_RASTER_LOGIT_OPACITY, f->xys, f->radii, f->conics, f->opacities, f->cum_tiles_hit,
                              &f->cam, f->channels == 4 ? f->depths : nullptr, f->splats, stream));
    const float* tight = (f->flags & TS_FRAME_TIGHT) ? f->splats : nullptr;
    TS_TRY(ts_bin_count(f->n, f->xys, f->radii, tight, &f->cam, f->bin_ws, stream));
    // (TS_FRAME_LIST_STATS: the longest list goes to the word behind the count word - read a frame later by the caller's
    // launch policy, never waited for)
    int32_t* longest = nullptr;
    if ((f->flags & TS_FRAME_LIST_STATS) && f->total_host) {
        int32_t* dev = mapped_pointer(f->total_host);
        if (dev) longest = dev + 1;
    }
    TS_TRY(ts_tile_offsets_stats(f->n, num_tiles(f), f->bin_ws, f->tile_bins, f->cum_tiles_hit, f->capacity, longest, stream));
    return 0;
}

int ts_frame_fwd_composite(const ts_frame* f, void* stream) {
    TsRange range_("ts_frame_fwd_composite");
    if (bad(f) || f->num_intersects < 0) return TS_E_BADARG;
    const bool planes = (f->flags & TS_FRAME_PLANES) != 0;
    if (planes && (f->channels != 4 || !f->out_depth)) return TS_E_BADARG;
    // 16x16 lists, one wave or (split) one workgroup per tile: the compositing kernel sorts the lists of <= 1024
    // entries itself
    const bool fused_sort = f->num_intersects > 0 && f->cam.wide_tiles == 0 &&
                            !(f->flags & (TS_FRAME_NARROW_WAVES | TS_FRAME_SEPARATE_SORT));
    if (f->num_intersects > 0) {
        const float* tight = (f->flags & TS_FRAME_TIGHT) ? f->splats : nullptr;
        // the sorted-id buffer is dead until the sort: it carries the ids between the two scatter hops
        TS_TRY(ts_bin_scatter(f->n, f->xys, f->radii, tight, &f->cam, f->bin_ws, f->bucket_ids,
                              (f->flags & TS_FRAME_DIRECT_SCATTER) ? nullptr : f->gaussian_ids_sorted, stream));
        int32_t* counter = f->bin_ws + (ts_bin_ws_ints(f->n, num_tiles(f)) - 1);
        if (fused_sort)
            TS_TRY(ts_sort_tiles_above(num_tiles(f), f->tile_bins, f->depths, f->bucket_ids, f->gaussian_ids_sorted,
                                       f->bin_ws, counter, stream));
        else
            TS_TRY(ts_sort_tiles(num_tiles(f), f->tile_bins, f->depths, f->bucket_ids, f->gaussian_ids_sorted,
                                 f->bin_ws, counter, stream));
    }
    if (fused_sort)
        return ts_raster_fwd_sort(f->channels, raster_flags(f), &f->cam, f->tile_bins, f->bucket_ids, f->depths,
                                  f->gaussian_ids_sorted, f->splats, f->background, f->out_img,
                                  planes ? f->out_depth : nullptr, f->final_Ts, f->final_index, f->clamp_mask, stream);
    return ts_raster_fwd_planes(f->channels, raster_flags(f), &f->cam, f->tile_bins, f->gaussian_ids_sorted, f->splats,
                                f->background, f->out_img, planes ? f->out_depth : nullptr, f->final_Ts,
                                f->final_index, f->clamp_mask, stream);
}

int ts_frame_bwd_composite(const ts_frame* f, void* stream) {
    TsRange range_("ts_frame_bwd_composite");
    if (bad(f) || f->num_intersects < 0) return TS_E_BADARG;
    const bool planes = (f->flags & TS_FRAME_PLANES) != 0;
    TS_TRY(ts_raster_bwd_planes(f->channels,
                                ((raster_flags(f) & ~(TS_RASTER_CLAMP_RGB | TS_RASTER_SPLIT_BLOCKS)) | bwd_split(f)) |
                                    TS_RASTER_FLAG_GEN(f->flag_gen),
                                f->num_intersects, &f->cam, f->tile_bins, f->gaussian_ids_sorted, f->splats,
                                f->background, f->final_Ts, f->final_index, f->v_out_img,
                                planes ? f->v_out_depth : nullptr, planes ? 1 : 0, nullptr, f->clamp_mask, f->partials,
                                f->row_flags, stream));
    const bool stripe = (f->flags & TS_FRAME_STRIPE) != 0;
    return ts_reduce_partials(f->n, f->channels,
                              TS_RASTER_LOGIT_OPACITY | bwd_split(f) | TS_RASTER_FLAG_GEN(f->flag_gen),
                              f->num_tiles_hit, f->cum_tiles_hit, f->partials, f->row_flags, f->splats, f->v_xy,
                              f->v_conic, f->v_colors, f->v_opacity, f->channels == 4 ? f->v_depth : nullptr,
                              stripe ? f->sh_mask : nullptr, stream);
}

int ts_frame_bwd_params(const ts_frame* f, void* stream) {
    TsRange range_("ts_frame_bwd_params");
    if (bad(f)) return TS_E_BADARG;
    TS_TRY(ts_sh_colors_bwd(f->n, f->sh_degree, f->num_bases, f->means, f->origin,
                            (f->flags & TS_FRAME_STRIPE) ? nullptr : f->sh_mask, f->v_colors,
                            f->v_colors_dc, f->num_bases > 1 ? f->v_colors_rest : nullptr, stream));
    return ts_project_bwd(f->n, f->means, f->scales, f->quats, f->view34, f->projview, &f->cam, 3, f->radii,
                          f->v_xy, f->v_depth, f->v_conic, nullptr, f->v_means, f->v_scales, f->v_quats, stream);
}

int ts_frame_bwd_params_adam(const ts_frame* f, const ts_adam* adam, void* stream) {
    TsRange range_("ts_frame_bwd_params_adam");
    if (bad(f) || !adam) return TS_E_BADARG;
    // (the colour stage reads the means for its view directions: it runs BEFORE the projection's pass moves them)
    TS_TRY(ts_sh_colors_bwd_adam(f->n, f->sh_degree, f->num_bases, f->means, f->origin,
                                 (f->flags & TS_FRAME_STRIPE) ? nullptr : f->sh_mask, f->v_colors,
                                 const_cast<float*>(f->colors_dc), f->num_bases > 1 ? const_cast<float*>(f->colors_rest) : nullptr,
                                 adam, stream));
    return ts_project_bwd_adam(f->n, const_cast<float*>(f->means), const_cast<float*>(f->scales), const_cast<float*>(f->quats),
                               f->view34, f->projview, &f->cam, 3, f->radii, f->v_xy, f->v_depth, f->v_conic,
                               const_cast<float*>(f->opacities), f->v_opacity, adam, stream);
}

// ---- Gaussian-sharded frame (csrc/shard.hip, tinysplat_amd/sharded.py): the same executor idea -----------------
// Two ts_frame describe a rank's frame: `fo` its OWNED Gaussians with the full-frame camera, `fs` the records its
// stripe imported (n = records, cam = the stripe); the per-stage entries are called in the order sharded.py
// documents.  ts_frame_fwd_composite(fs) is the stripe's scatter + sort + compositing, unchanged.

int ts_shard_owner_fwd(const ts_frame* fo, const ts_stripes* stripes, int32_t* route_ws, int32_t* counts,
                       void* stream) {
    return ts_shard_owner_fwd_padded(fo, stripes, nullptr, route_ws, counts, stream);
}

int ts_shard_owner_fwd_padded(const ts_frame* fo, const ts_stripes* stripes, const int32_t* group_base,
                              int32_t* route_ws, int32_t* counts, void* stream) {
    TsRange range_("ts_shard_owner_fwd");
    if (bad(fo) || !stripes) return TS_E_BADARG;
    if (fo->n > 0 && fo->n <= small_n_fused())          // a small shard: one launch instead of three (shard.hip)
        return ts_shard_owner_fwd_fused(fo->n, fo->sh_degree, fo->num_bases, fo->means, fo->scales, fo->quats, fo->view34,
                                        fo->projview, &fo->cam, 3, fo->origin, fo->colors_dc,
                                        fo->num_bases > 1 ? fo->colors_rest : nullptr, fo->opacities, fo->channels,
                                        TS_RASTER_LOGIT_OPACITY, fo->xys, fo->depths, fo->radii, fo->conics,
                                        fo->num_tiles_hit, fo->sh_mask, fo->splats, stripes, group_base, route_ws, counts,
                                        stream);
    TS_TRY(ts_project_fwd(fo->n, fo->means, fo->scales, fo->quats, fo->view34, fo->projview, &fo->cam, 3, fo->xys,
                          fo->depths, fo->radii, fo->conics, fo->num_tiles_hit, nullptr, stream));
    // colour stage + packed records of the owned Gaussians; the slot fields are rewritten by the importing rank,
    // so any int array serves as cum_tiles_hit
    TS_TRY(ts_colors_pack_fwd(fo->n, fo->sh_degree, fo->num_bases, fo->means, fo->origin, fo->colors_dc,
                              fo->num_bases > 1 ? fo->colors_rest : nullptr, fo->sh_mask, nullptr, fo->channels,
                              TS_RASTER_LOGIT_OPACITY, fo->xys, fo->radii, fo->conics, fo->opacities,
                              fo->num_tiles_hit, &fo->cam, fo->channels == 4 ? fo->depths : nullptr, fo->splats,
                              stream));
    return ts_route_count_padded(fo->n, fo->xys, fo->radii, &fo->cam, stripes, group_base, route_ws, counts, stream);
}

int ts_shard_stripe_fwd_import(const ts_frame* fs, const float* records, void* stream) {
    TsRange range_("ts_shard_stripe_fwd_import");
    if (bad(fs)) return TS_E_BADARG;
    if (fs->n > 0) {
        TS_TRY(ts_import_records(fs->n, records, &fs->cam, fs->xys, fs->depths, fs->radii, fs->num_tiles_hit, stream));
        int32_t* total_dev = fs->total_host ? mapped_pointer(fs->total_host) : nullptr;
        TS_TRY(ts_scan_tiles(fs->n, fs->num_tiles_hit, fs->cum_tiles_hit, fs->scan_ws, total_dev, stream));
        if (fs->total_host && !total_dev) {
            const hipError_t e = hipMemcpyAsync(fs->total_host, fs->cum_tiles_hit + (fs->n - 1), sizeof(int32_t),
                                                hipMemcpyDeviceToHost, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
        TS_TRY(ts_import_pack(fs->n, records, fs->cum_tiles_hit, &fs->cam, fs->splats, stream));
    }
    const float* tight = (fs->flags & TS_FRAME_TIGHT) ? fs->splats : nullptr;
    TS_TRY(ts_bin_count(fs->n, fs->xys, fs->radii, tight, &fs->cam, fs->bin_ws, stream));
    return ts_tile_offsets(fs->n, num_tiles(fs), fs->bin_ws, fs->tile_bins, fs->cum_tiles_hit, fs->capacity, stream);
}

int ts_shard_stripe_bwd(const ts_frame* fs, float* grad_rows, void* stream) {
    TsRange range_("ts_shard_stripe_bwd");
    if (bad(fs) || fs->num_intersects < 0) return TS_E_BADARG;
    TS_TRY(ts_raster_bwd(fs->channels,
                         ((raster_flags(fs) & ~(TS_RASTER_CLAMP_RGB | TS_RASTER_SPLIT_BLOCKS)) | bwd_split(fs)) |
                             TS_RASTER_FLAG_GEN(fs->flag_gen),
                         fs->num_intersects, &fs->cam, fs->tile_bins, fs->gaussian_ids_sorted, fs->splats,
                         fs->background, fs->final_Ts, fs->final_index, fs->v_out_img, nullptr, fs->clamp_mask,
                         fs->partials, fs->row_flags, stream));
    return ts_reduce_partials_rows(fs->n, fs->channels,
                                   bwd_split(fs) | TS_RASTER_FLAG_GEN(fs->flag_gen),
                                   fs->num_tiles_hit, fs->cum_tiles_hit, fs->partials, fs->row_flags, fs->splats,
                                   grad_rows, stream);
}

int ts_shard_owner_bwd(const ts_frame* fo, const ts_stripes* stripes, const int32_t* route_ws,
                       const float* grad_rows, void* stream) {
    TsRange range_("ts_shard_owner_bwd");
    if (bad(fo) || !stripes) return TS_E_BADARG;
    if (fo->n > 0 && fo->n <= small_n_fused())          // a small shard: one launch instead of three (shard.hip)
        return ts_shard_owner_bwd_fused(fo->n, fo->channels, fo->sh_degree, fo->num_bases, fo->means, fo->scales, fo->quats,
                                        fo->view34, fo->projview, fo->origin, fo->xys, fo->radii, fo->splats, fo->sh_mask,
                                        &fo->cam, stripes, route_ws, grad_rows, fo->v_xy, fo->v_conic, fo->v_colors,
                                        fo->channels == 4 ? fo->v_depth : nullptr, fo->v_opacity, fo->v_colors_dc,
                                        fo->num_bases > 1 ? fo->v_colors_rest : nullptr, fo->v_means, fo->v_scales,
                                        fo->v_quats, stream);
    TS_TRY(ts_route_accumulate(fo->n, fo->channels, fo->xys, fo->radii, fo->splats, fo->sh_mask, &fo->cam, stripes,
                               route_ws, grad_rows, fo->v_xy, fo->v_conic, fo->v_colors,
                               fo->channels == 4 ? fo->v_depth : nullptr, fo->v_opacity, stream));
    TS_TRY(ts_sh_colors_bwd(fo->n, fo->sh_degree, fo->num_bases, fo->means, fo->origin, nullptr, fo->v_colors,
                            fo->v_colors_dc, fo->num_bases > 1 ? fo->v_colors_rest : nullptr, stream));
    return ts_project_bwd(fo->n, fo->means, fo->scales, fo->quats, fo->view34, fo->projview, &fo->cam, 3, fo->radii,
                          fo->v_xy, fo->channels == 4 ? fo->v_depth : nullptr, fo->v_conic, nullptr, fo->v_means,
                          fo->v_scales, fo->v_quats, stream);
}

// ---- the whole step of one rank in four calls (ts_rank_step): the entries above composed, nothing new -------------
int32_t ts_rank_step_struct_bytes(void) { return (int32_t)sizeof(ts_rank_step); }

int ts_shard_rank_fwd_a(const ts_frame* fo, const ts_rank_step* r, void* stream) {
    if (bad(fo) || !r || r->send_rows < 0 || r->stripes.num < 1 || r->stripes.num > TS_MAX_RANKS) return TS_E_BADARG;
    if (r->group_base[r->stripes.num] != r->send_rows || (r->send_rows > 0 && !r->send)) return TS_E_BADARG;
    TS_TRY(ts_shard_owner_fwd_padded(fo, &r->stripes, r->group_base, r->route_ws, r->counts, stream));
    if (r->send_rows > 0) {
        // an all-zero record lists nothing at its destination (radius 0): the padding of every group
        const hipError_t e = hipMemsetAsync(r->send, 0, (size_t)r->send_rows * TS_EXPORT_RECORD_FLOATS * sizeof(float),
                                            (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    TsRange range_("ts_route_pack");
    return ts_route_pack(fo->n, r->gid_base, fo->xys, fo->radii, fo->depths, fo->splats, &fo->cam, &r->stripes,
                         r->route_ws, r->send, stream);
}

int ts_shard_rank_fwd_b(const ts_frame* fs, const ts_rank_step* r, void* stream) {
    if (bad(fs) || !r || fs->n != r->recv_rows || fs->capacity < 0) return TS_E_BADARG;
    TS_TRY(ts_shard_stripe_fwd_import(fs, fs->n > 0 ? r->recv : nullptr, stream));
    return ts_frame_fwd_composite(fs, stream);
}

int ts_shard_rank_bwd_a(const ts_frame* fs, const ts_rank_step* r, void* stream) {
    if (!r) return TS_E_BADARG;
    return ts_shard_stripe_bwd(fs, r->grad_rows, stream);
}

int ts_shard_rank_bwd_b(const ts_frame* fo, const ts_rank_step* r, void* stream) {
    if (!r) return TS_E_BADARG;
    return ts_shard_owner_bwd(fo, &r->stripes, r->route_ws, r->back, stream);
}

}  // extern "C"
