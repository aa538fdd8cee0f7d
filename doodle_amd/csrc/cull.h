// Rays whose footprint is EXACTLY zero on a tile — found before the footprint kernels run, skipped by them.
//
// The reference evaluates exp(-|P_ij - x|²/2σ²) for every (ray, pixel) pair
// (newenv_rl_test_multi_error.py:142-148) and so do the dense kernels here.  With orientation errors of
// tens of mrad most reflections miss the receiver by many σ: their factors underflow and every fused
// multiply-add they take part in leaves its accumulator unchanged.  Leaving such a ray out of a sum is
// therefore bit-identical — provided "zero" is decided from what the kernels COMPUTE, not from the real
// number.  The kernels form a factor in one of three ways (coordinate v, shift s = a or b, cc = c2 or 0):
//     q-form    exp2(-fma(q, q, cc·k2)),  q = fma(v, √k2, s·√k2)        (MFMA kernels; √k2 correctly rounded)
//     fused     exp2(-(fma(t, t, cc) · k2)),  t = v + s                 (VALU kernels)
//     unfused   exp2( ((t·t) + cc) · (-k2) )                            (epilogue of the MFMA backward)
// Each is a composition of correctly rounded — hence monotone — operations of |q| or |t|, so the smallest
// exponent over a tile's coordinates [lo, hi] (the true min and max of the tile's xs / ys: no ordering of
// the coordinate arrays is assumed) is the same expression evaluated at the end nearer to -s, or 0 when
// the interval straddles it.  exponent_floor() is the least of the three; a NaN anywhere — in a ray's parameters
// or in the tile's coordinates, on EITHER axis — keeps the ray (the dense kernels turn 0·NaN into NaN).
// tests/c/cull_floor.cpp checks all of this on the CPU, bit for bit, by brute force (cull_math.h).
//   * forward: a ray is dropped from a tile when floor_x + floor_y > CULL_EXP2: every product A_i·E_j is
//     then below 2^-152 (the margin of 2 covers v_exp_f32's error and its handling of denormal results, if
//     it flushes them the product is 0 outright), less than half an ulp of ANY f32 accumulator, and
//     fma(A_i, E_j, acc) == acc.
//   * backward: a ray is dropped from an image when floor_x > CULL_EXP2 or floor_y > CULL_EXP2 over the WHOLE
//     image: one factor table is then all +0, both contractions and all five moments of the dense kernels
//     are exactly +0, and zeros are what the compaction kernel writes for it.  Where an image is several c tiles
//     of the LDS-tile kernels wide (R > 256) the same holds per (pass, c tile): pass 0's moments of a tile of
//     COLUMNS are Σ_j E_j·w_j·(Σ_i G_ij A_i) over that tile's columns and all rows — exactly +0 when the tile's
//     E_j are all +0 or the image's A_i are (finite cotangent), so the ray leaves that tile's list when
//     floor_y(tile) > CULL_EXP2 or floor_x(image) > CULL_EXP2; pass 1 likewise with rows and columns exchanged.
// A plane-parallel ray (k2 = 0) has exponent 0 everywhere and is always kept: it adds 1.0 to every pixel
// (:141-148).
#pragma once
#include <hip/hip_runtime.h>

#define HELIO_HD __device__ __forceinline__
#include "cull_math.h"
#undef HELIO_HD
namespace helio {

__device__ __forceinline__ bool cull_dead_product(const float4 ray, const CullBox& bx) {
    return cull_dead_product(ray.x, ray.y, ray.z, ray.w, bx);
}
__device__ __forceinline__ bool cull_dead_strict(const float4 ray, const CullBox& bx) {
    return cull_dead_strict(ray.x, ray.y, ray.z, ray.w, bx);
}

// ---- work order ---------------------------------------------------------------------------------------
// The chip hands workgroups to its 8 XCDs round-robin by linear id and in order: when consecutive ids carry
// unequal work the dispatcher waits on the busiest XCD and the others idle (measured: a compacted backward
// whose empty tiles were interleaved with the full ones took as long as the dense one).  So the kernels do
// not take their tile from blockIdx directly but from a small table made with the lists:
//   forward  — order[w] (made when there are more lists than the chip has CUs — below that every workgroup starts
//              at once and the table's launch is only latency): the (image, tile) lists sorted by length, longest first (neighbouring ids: nearly
//              equal work; long ones early: a short tail);
//   backward — map[w]: the (image, 256-ray tile) items that are not empty, image-major (equal work each), and
//              their number; ids past it leave at once, all at the END of the grid.
// Which workgroup computes a tile does not enter any result.
//
// ---- scratch layouts (device memory handed in by the caller; helio_*_scratch_bytes) ----------------
// forward:  int counts[T] | int order[T] | float4 lists[T][P]          T = B·tiles²·S, tile = TE×TE pixels;
//           S = 1, P = N normally; with the heliostat sum split across workgroups (splat_fwd.hip, "split"),
//           S parts of P consecutive rays each and one list per (image, tile, part)
// backward: int counts[T] | int idx[T][N] | int total[sets], tail_total[sets] (256 bytes) | int2 map[T·⌈N/tile_rays⌉] |
//           int2 tail_map[T]     T = sets·B·CT lists:
//           CT = 1: one list per image, shared by the two passes (sets = 1); CT > 1: one list per (pass, image,
//           c tile) of the LDS-tile kernels (sets = 2), list = (pass·B + b)·CT + tile.  tail_map: the LAST tile of a
//           list when it holds at most 128 rays — those run in the kernel's 128-ray form (splat_bwd.hip), the others
//           (map) in the 256-ray form.
// (every section padded to 256 bytes)
__host__ __device__ inline long cull_pad256(long bytes) { return (bytes + 255) & ~255l; }
inline long cull_fwd_bytes(int B, int N, int R, int TE, int S = 1, int P = 0) {
    const long t = (R + TE - 1) / TE, T = (long)B * t * t * S;
    return 2 * cull_pad256(4 * T) + 16 * T * (S > 1 ? P : N);
}
constexpr int CULL_BWD_TILE = 256;           // rays per tile of splat_bwd_mfma (64 in its 64-ray form: the map counts tiles of `tile_rays`)
constexpr int CULL_BWD_MAX_CT = 8;           // c tiles per image that get lists of their own (R ≤ 2048)
inline long cull_bwd_lists(int B, int CT) { return (long)B * CT * (CT > 1 ? 2 : 1); }
inline long cull_bwd_bytes(int B, int N, int CT = 1, int tile_rays = CULL_BWD_TILE) {
    const long nt = (N + tile_rays - 1) / tile_rays, T = cull_bwd_lists(B, CT);
    return cull_pad256(4 * T) + cull_pad256(4 * T * N) + 256 + 8 * T * nt + 8 * T;
}

struct CullFwd { const int* counts; const int* order; const float4* lists; };      // counts == nullptr: dense
// ct: lists per image and pass (1 = one per image for both passes); for_pass(): the set a pass walks
struct CullBwd {
    const int* counts; const int* idx; const int* total; const int2* map; const int* tail_total; const int2* tail_map;
    int ct; long set_lists, set_items; int N;
    CullBwd for_pass(int pass) const {
        if (!counts || ct <= 1 || pass == 0) return *this;
        return CullBwd{counts + set_lists, idx + set_lists * N, total + 1, map + set_items, tail_total + 1, tail_map + set_lists,
                       ct, set_lists, set_items, N};
    }
};

// launchers (cull.hip)
CullFwd launch_cull_fwd(int B, int N, int R, int TE, int S, int P, bool with_order, const float* rays, const float* xs,
                        const float* ys, void* scratch, hipStream_t st);
// TC: width of a c tile in pixels, CT = ⌈R/TC⌉ or 1; tile_rays: rays per item of the map (256; 64 for the 64-ray tiles, no tails)
CullBwd launch_cull_bwd(int B, int N, int R, int JB, int TC, int CT, bool with_map, bool split_tails, const float* rays, const float* xs,
                        const float* ys, float* moments, void* scratch, hipStream_t st, int tile_rays = CULL_BWD_TILE);
bool cull_enabled();

}  // namespace helio
