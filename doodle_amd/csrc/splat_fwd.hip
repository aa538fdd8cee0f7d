// Forward Gaussian-footprint accumulation ("splat") for gfx950.
//
// Replaces gaussian_blur_batch + sum over heliostats of the reference
// (newenv_rl_test_multi_error.py :107-149, :404-406).  The reference evaluates
// exp(-|P_ij - x|²/2σ²) for every (ray, pixel); because plane_u ⟂ plane_v and
// both are unit (:206-213) the footprint factorises exactly,
//     gauss_bn[i,j] = A_bn[i] · E_bn[j],
//     A_bn[i] = exp2(-((xs[i]+a)² + c2)·k2),   E_bn[j] = exp2(-(ys[j]+b)²·k2),
// so an image is a rank-N sum of outer products: 2·R exps and R² FMAs per ray.
// It is a gather (every pixel visits every ray): one workgroup owns an output
// tile, no atomics, deterministic.
//
// Two kernels compute that sum:
//   * splat_fwd_valu — the factors of a chunk of rays are staged in LDS, every
//     thread keeps an 8×8 (or 4×4) register tile of pixels and does FMAs.
//   * splat_fwd_mfma — the outer-product sum is issued on the matrix pipe with
//     v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: an exact fmaf chain, same
//     numerics as the VALU kernel); every lane computes its own A/E operand in
//     registers, so the VALU only generates exponentials while the matrix pipe
//     does the accumulation.  No LDS traffic in the inner loop.
// Both accumulate in two levels (a chunk of rays, then the running total) so that
// the rounding error of the sum over N stays at the cascade-sum level of torch.
#include <hip/hip_runtime.h>
#include "helio.h"

namespace helio {

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// ----------------------------------------------------------------------------------------------
// VALU variant
// ----------------------------------------------------------------------------------------------
template <int TILE, int NC>
__global__ void __launch_bounds__(256)
splat_fwd_valu(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
               const float* __restrict__ ys, float* __restrict__ image) {
    constexpr int TR = TILE / 16;        // rows per thread
    constexpr int CG = TILE / 64;        // groups of 4 columns per thread
    constexpr int TC = 4 * CG;
    constexpr int PER = NC * TILE / 256; // factors of each kind a thread stages per chunk
    __shared__ __attribute__((aligned(16))) float sA[NC][TILE];
    __shared__ __attribute__((aligned(16))) float sE[NC][TILE];

    const int tiles = (R + TILE - 1) / TILE;
    const int b = blockIdx.y;
    const int i0 = (blockIdx.x / tiles) * TILE, j0 = (blockIdx.x % tiles) * TILE;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;

    // staging role: this thread always produces column p of the factor tables
    const int p = tid % TILE, nn0 = tid / TILE;
    const float xp = xs[min(i0 + p, R - 1)], yp = ys[min(j0 + p, R - 1)];
    const float4* __restrict__ rb = reinterpret_cast<const float4*>(rays) + (long)b * N;

    float tot[TR][TC], acc[TR][TC];
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c) tot[r][c] = 0.0f;

    for (int n0 = 0; n0 < N; n0 += NC) {
        __syncthreads();   // previous chunk fully consumed
#pragma unroll
        for (int s = 0; s < PER; ++s) {
            const int nn = nn0 + s * (256 / TILE);
            const int n = n0 + nn;
            float fa = 0.0f, fe = 0.0f;
            if (n < N) {
                const float4 q = rb[n];            // (a, b, k2, c2); broadcast within the wave
                const float t = xp + q.x, u = yp + q.y;
                fa = exp2_fast(-(__builtin_fmaf(t, t, q.w) * q.z));
                fe = exp2_fast(-((u * u) * q.z));
            }
            sA[nn][p] = fa;
            sE[nn][p] = fe;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int c = 0; c < TC; ++c) acc[r][c] = 0.0f;
#pragma unroll 4
        for (int nn = 0; nn < NC; ++nn) {
            float av[TR], ev[TC];
#pragma unroll
            for (int r4 = 0; r4 < TR / 4; ++r4) {
                const float4 v = *reinterpret_cast<const float4*>(&sA[nn][ty * TR + 4 * r4]);
                av[4 * r4] = v.x; av[4 * r4 + 1] = v.y; av[4 * r4 + 2] = v.z; av[4 * r4 + 3] = v.w;
            }
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(&sE[nn][64 * g + 4 * tx]);
                ev[4 * g] = v.x; ev[4 * g + 1] = v.y; ev[4 * g + 2] = v.z; ev[4 * g + 3] = v.w;
            }
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int c = 0; c < TC; ++c) acc[r][c] = __builtin_fmaf(av[r], ev[c], acc[r][c]);
        }
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int c = 0; c < TC; ++c) tot[r][c] += acc[r][c];
    }

    float* __restrict__ img = image + (long)b * R * R;
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const int i = i0 + ty * TR + r;
        if (i >= R) continue;
#pragma unroll
        for (int g = 0; g < CG; ++g) {
            const int j = j0 + 64 * g + 4 * tx;
            float* dst = img + (long)i * R + j;
            if (j + 3 < R && (R & 3) == 0) {
                *reinterpret_cast<float4*>(dst) = make_float4(tot[r][4 * g], tot[r][4 * g + 1], tot[r][4 * g + 2], tot[r][4 * g + 3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (j + k < R) dst[k] = tot[r][4 * g + k];
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// f32 MFMA variant
// ----------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A workgroup = 4 waves = one 128×128 tile (WT=64: each wave a 64×64 quadrant, 2×2 MFMA
// blocks) or one 64×64 tile (WT=32: each wave one 32×32 block).  MFMA operand maps
// (cdna_hip_programming.md §3): lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// the two k of one instruction are two consecutive heliostats.
template <int WT, int NC>
__global__ void __launch_bounds__(256)
splat_fwd_mfma(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
               const float* __restrict__ ys, float* __restrict__ image) {
    constexpr int MB = WT / 32;          // MFMA blocks per wave along each axis
    constexpr int TILE = 2 * WT;
    __shared__ float4 sRay[NC];

    const int tiles = (R + TILE - 1) / TILE;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int i0 = (blockIdx.x / tiles) * TILE + (wave >> 1) * WT;
    const int j0 = (blockIdx.x % tiles) * TILE + (wave & 1) * WT;

    float xv[MB], yv[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        xv[m] = xs[min(i0 + 32 * m + lr, R - 1)];
        yv[m] = ys[min(j0 + 32 * m + lr, R - 1)];
    }
    f32x16 tot[MB][MB], acc[MB][MB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int mj = 0; mj < MB; ++mj)
#pragma unroll
            for (int e = 0; e < 16; ++e) tot[mi][mj][e] = 0.0f;

    const float4* __restrict__ rb = reinterpret_cast<const float4*>(rays) + (long)b * N;
    for (int n0 = 0; n0 < N; n0 += NC) {
        __syncthreads();
        for (int k = tid; k < NC; k += 256)
            sRay[k] = (n0 + k < N) ? rb[n0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int mj = 0; mj < MB; ++mj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][mj][e] = 0.0f;
        const int cnt = min(NC, N - n0);
#pragma unroll 2
        for (int k = 0; k < cnt; k += 2) {
            const float4 q = sRay[k + lh];
            const float live = (k + lh < cnt) ? 1.0f : 0.0f;   // odd tail: the missing ray adds 0
            float fa[MB], fe[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const float t = xv[m] + q.x, u = yv[m] + q.y;
                fa[m] = live * exp2_fast(-(__builtin_fmaf(t, t, q.w) * q.z));
                fe[m] = exp2_fast(-((u * u) * q.z));
            }
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int mj = 0; mj < MB; ++mj)
                    acc[mi][mj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi], fe[mj], acc[mi][mj], 0, 0, 0);
        }
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int mj = 0; mj < MB; ++mj) tot[mi][mj] += acc[mi][mj];
    }

    // C/D map of the 32×32 MFMA: column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* __restrict__ img = image + (long)b * R * R;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int mj = 0; mj < MB; ++mj) {
            const int j = j0 + 32 * mj + lr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (i < R && j < R) img[(long)i * R + j] = tot[mi][mj][e];
            }
        }
}

int launch_splat_fwd(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                     float* image, int variant, hipStream_t st) {
    if (variant == 0) variant = 2;
    // Small problems (few, small images) want more, smaller workgroups to fill 256 CUs.
    const long big_tiles = (long)B * ((R + 127) / 128) * ((R + 127) / 128);
    const bool small = big_tiles < 512;
    if (variant == 1) {
        if (small) {
            const int t = (R + 63) / 64;
            hipLaunchKernelGGL((splat_fwd_valu<64, 32>), dim3(t * t, B), dim3(256), 0, st, B, N, R, rays, xs, ys, image);
        } else {
            const int t = (R + 127) / 128;
            hipLaunchKernelGGL((splat_fwd_valu<128, 32>), dim3(t * t, B), dim3(256), 0, st, B, N, R, rays, xs, ys, image);
        }
    } else if (variant == 2) {
        if (small) {
            const int t = (R + 63) / 64;
            hipLaunchKernelGGL((splat_fwd_mfma<32, 128>), dim3(t * t, B), dim3(256), 0, st, B, N, R, rays, xs, ys, image);
        } else {
            const int t = (R + 127) / 128;
            hipLaunchKernelGGL((splat_fwd_mfma<64, 128>), dim3(t * t, B), dim3(256), 0, st, B, N, R, rays, xs, ys, image);
        }
    } else {
        return HELIO_E_INVALID;
    }
    return HELIO_OK;
}

}  // namespace helio
