// Sorting networks of the per-tile sort (device code shared by binning.hip - the sort kernels - and raster.hip -
// the forward compositing kernel that sorts a tile's list itself before it composites it).  Included inside the
// including file's anonymous namespace.
#pragma once

// sort key of a bucket entry: depth bits (positive floats order like their bit patterns) then id
__device__ __forceinline__ unsigned long long make_key(const float* __restrict__ depths, int id) {
    return ((unsigned long long)__float_as_uint(depths[id]) << 32) | (unsigned int)id;
}


// Register-resident bitonic sort of one tile bucket by a 256-thread workgroup: thread t holds the E
// consecutive keys t*E .. t*E+E-1 (padded with +inf to npad = 256*E).  Compare-exchange partners at
// distance j are in the same thread (j < E: pure register work), in the same wave (E <= j < 64 E:
// 64-bit lane exchange through the LDS crossbar, no barrier) or in another wave (j >= 64 E: one LDS
// round trip with a barrier - at most 3 of the 55 stages of a 1024-key sort).
// Bitonic network over GROUP threads (GROUP = 256: the workgroup, with LDS + barriers for the
// cross-wave stages; GROUP = 64: one wave, shuffles only, no barrier).  Thread t of the group holds
// the E consecutive keys t*E .. t*E+E-1; npad (a power of two <= GROUP*E) bounds the stages that can
// see anything but +inf padding.
// The 32-bit value of lane ^ M for M = 1, 2, 4, 8 without the LDS crossbar: quad_perm, row_ror:8, and for M = 4
// row_half_mirror (lane 7 - i of each 8) followed by a reversed quad_perm.  Every lane is written, so no previous
// value of the destination is needed (old = 0 with bound_ctrl: no copy in front of the DPP move).  26 of the 33
// in-wave stages of a 2048-key sort exchange at these distances; with ds_bpermute for all of them the LDS pipe
// was the busiest unit of the sort (config 5: ~675 LDS instructions per wave, SQ_ACTIVE_INST_LDS at the kernel's
// duration).
template <int M>
__device__ __forceinline__ unsigned int lane_xor_dpp(unsigned int x) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "DPP reaches lane ^ 1, 2, 4, 8");
    const int v = (int)x;
    if constexpr (M == 1) return (unsigned int)__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    if constexpr (M == 2) return (unsigned int)__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    if constexpr (M == 8) return (unsigned int)__builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);   // row_ror:8
    if constexpr (M == 4) {
        const int r = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);                            // row_half_mirror
        return (unsigned int)__builtin_amdgcn_update_dpp(0, r, 0x1B, 0xF, 0xF, true);                      // quad_perm:[3,2,1,0]
    }
    return x;
}

#ifndef TS_SORT_WAVE_DPP
#define TS_SORT_WAVE_DPP 0          // 1: DPP exchanges in the one-wave-per-tile sort too (A/B knob)
#endif
template <int E, int M>
__device__ __forceinline__ void exchange_stage_dpp(unsigned long long (&k)[E], bool keep_min) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned long long o = ((unsigned long long)lane_xor_dpp<M>((unsigned int)(k[e] >> 32)) << 32) |
                                     lane_xor_dpp<M>((unsigned int)k[e]);
        k[e] = ((o < k[e]) == keep_min) ? o : k[e];                   // keys are unique: no ties
    }
}

template <int E, int GROUP>
__device__ __forceinline__ void bitonic_regs(unsigned long long (&k)[E], int t, int npad,
                                             unsigned long long* lds) {
    // lane ^ 1, 2, 4, 8 by DPP where the LDS pipe is the busy unit (a workgroup per tile: four waves per tile and
    // five and more tiles per CU); a wave sorting a tile on its own has that pipe to spare and fewer VALU
    // instructions with ds_bpermute (config 3's sort: 75 us, 80 with DPP)
    constexpr bool kDpp = TS_SORT_WAVE_DPP || GROUP > 64;
    const int lane = t & 63;
    for (int kk = 2; kk <= npad; kk <<= 1) {
        for (int j = kk >> 1; j >= E && j >= 1; j >>= 1) {
            // for j >= E the direction bit (i & kk) and the side bit (i & j) depend on t only
            const bool asc = ((t * E) & kk) == 0;
            if (GROUP > 64 && j >= 64 * E) {         // partner in another wave: thread t ^ (j / E), same register
                // element e of thread t at lds[e * GROUP + t] (consecutive lanes, consecutive 8-byte words), half of
                // the elements per round: GROUP * E / 2 words of LDS, so that a 4096-key sort needs 16 KiB, not 32
                constexpr int H = E >= 2 ? E / 2 : 1;
                const bool keep_min = (((t * E) & j) == 0) == asc;
                const int partner = t ^ (j / E);
#pragma unroll
                for (int r = 0; r < E / H; ++r) {
#pragma unroll
                    for (int e = 0; e < H; ++e) lds[e * GROUP + t] = k[r * H + e];
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < H; ++e) {
                        const unsigned long long o = lds[e * GROUP + partner];
                        k[r * H + e] = ((o < k[r * H + e]) == keep_min) ? o : k[r * H + e];
                    }
                    __syncthreads();
                }
            } else {                                 // partner lane = lane ^ (j / E), same register
                const int m = j / E;
                const bool keep_min = ((lane & m) == 0) == asc;
                switch (kDpp ? m : 0) {
                    case 1: exchange_stage_dpp<E, 1>(k, keep_min); break;
                    case 2: exchange_stage_dpp<E, 2>(k, keep_min); break;
                    case 4: exchange_stage_dpp<E, 4>(k, keep_min); break;
                    case 8: exchange_stage_dpp<E, 8>(k, keep_min); break;
                    default:                         // lane ^ 16, lane ^ 32: through the LDS crossbar
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const unsigned long long o = __shfl_xor(k[e], m, 64);
                            k[e] = ((o < k[e]) == keep_min) ? o : k[e];
                        }
                }
            }
        }
#pragma unroll
        for (int jj = E >> 1; jj >= 1; jj >>= 1) {   // in-thread stages (compile-time register ids)
            if (jj < kk) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & jj) == 0) {
                        const bool asc = (((t * E + e) & kk) == 0);
                        const unsigned long long x = k[e], y = k[e | jj];
                        const bool sw = (x > y) == asc;
                        k[e] = sw ? y : x;
                        k[e | jj] = sw ? x : y;
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ int pow2_at_least(int n) {
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}


// One WAVE sorts one tile (n <= 64 E keys, E = 1..16 per lane): the whole bitonic network runs on
// registers and lane exchanges, no LDS storage, no barrier, and the E independent keys of a lane
// keep E exchanges in flight per stage.  Which key starts in which position is irrelevant to a sort,
// so the ids are loaded coalesced (position e*64 + lane); the result is stored by position.
#ifndef TS_WAVE_SORT_MAX
#define TS_WAVE_SORT_MAX 1024   // 512 / 256 (more tiles to the workgroup sort): config 3's sort 74 -> 84 / 94 us
#endif
constexpr int kWaveSortMax = TS_WAVE_SORT_MAX;
template <int E>
__device__ __forceinline__ void sort_tile_wave(const int* __restrict__ g,
                                               const float* __restrict__ depths,
                                               int* __restrict__ out, int n, int lane) {
    unsigned long long k[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        k[e] = i < n ? make_key(depths, g[i]) : ~0ull;
    }
    bitonic_regs<E, 64>(k, lane, min(pow2_at_least(n), 64 * E), nullptr);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        if (i < n) out[i] = (int)(unsigned int)k[e];
    }
}

