// Backward of the Gaussian-footprint accumulation for gfx950.
//
// Autograd of the reference's gaussian_blur_batch + sum (newenv_rl_test_multi_error.py
// :107-149, :404-406) w.r.t. the per-ray footprint parameters reduces to five centred
// moments per ray of  Gg = grad_image[b] ⊙ gauss_bn  (SURVEY.md §7.2):
//     M = Σ_ij Gg · (1, t_i, s_j, t_i², s_j²),   t_i = xs[i]+a,  s_j = ys[j]+b.
// With gauss = A[i]·E[j] they factor as
//     U_k[n,j] = Σ_i G[i,j] · t_i^k A_n[i]   (k = 0,1,2)      ← 3 FMAs per (ray, pixel)
//     M0 = Σ_j E U_0,  Ms = Σ_j s E U_0,  Mss = Σ_j s² E U_0,  Mt = Σ_j E U_1,  Mtt = Σ_j E U_2.
// A workgroup owns 64 rays × 64 image columns and sweeps all R rows: the weighted row
// factors of a 32-row chunk and the matching slab of grad_image are staged in LDS, every
// thread keeps a (4 rays × 3 × 4 columns) register tile, and the final reduction over
// columns is a 16-lane wavefront shuffle.  Column blocks write separate partials
// (moments[b, jb, n, :]) that helio_geometry_bwd adds in fixed order: no atomics,
// bit-reproducible.
#include <hip/hip_runtime.h>
#include "helio.h"

namespace helio {

constexpr int BW_NT = 64;   // rays per workgroup
constexpr int BW_JT = 64;   // image columns per workgroup
constexpr int BW_IC = 32;   // image rows per LDS chunk

__global__ void __launch_bounds__(256)
splat_bwd_valu(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
               const float* __restrict__ ys, const float* __restrict__ gimg, float* __restrict__ moments) {
    __shared__ __attribute__((aligned(16))) float sA[3][BW_IC][BW_NT];
    __shared__ __attribute__((aligned(16))) float sG[BW_IC][BW_JT];
    __shared__ float4 sRay[BW_NT];

    const int JB = (R + BW_JT - 1) / BW_JT;
    const int jb = blockIdx.x % JB, nb = blockIdx.x / JB, b = blockIdx.y;
    const int n0 = nb * BW_NT, j0 = jb * BW_JT;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;

    // staging role for the row factors: ray nl (fixed per thread), rows il0 + 4 s
    const int nl = tid & 63, il0 = tid >> 6;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool ray_ok = n0 + nl < N;
    if (ray_ok) q = reinterpret_cast<const float4*>(rays)[(long)b * N + n0 + nl];
    if (tid < BW_NT) sRay[tid] = q;

    float acc[4][3][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][k][c] = 0.0f;

    const float* __restrict__ G = gimg + (long)b * R * R;
    const bool vec_ok = (R & 3) == 0;
    for (int i0 = 0; i0 < R; i0 += BW_IC) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BW_IC / 4; ++s) {
            const int il = il0 + 4 * s, i = i0 + il;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            if (ray_ok && i < R) {
                const float t = xs[i] + q.x;
                a0 = __builtin_amdgcn_exp2f(-(__builtin_fmaf(t, t, q.w) * q.z));
                a1 = t * a0;
                a2 = t * a1;
            }
            sA[0][il][nl] = a0; sA[1][il][nl] = a1; sA[2][il][nl] = a2;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int idx = tid + 256 * s, j4 = idx & 15, il = idx >> 4;
            const int i = i0 + il, j = j0 + 4 * j4;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < R) {
                if (vec_ok && j + 3 < R) {
                    g = *reinterpret_cast<const float4*>(G + (long)i * R + j);
                } else {
                    if (j < R) g.x = G[(long)i * R + j];
                    if (j + 1 < R) g.y = G[(long)i * R + j + 1];
                    if (j + 2 < R) g.z = G[(long)i * R + j + 2];
                    if (j + 3 < R) g.w = G[(long)i * R + j + 3];
                }
            }
            *reinterpret_cast<float4*>(&sG[il][4 * j4]) = g;
        }
        __syncthreads();
#pragma unroll 4
        for (int il = 0; il < BW_IC; ++il) {
            const float4 g = *reinterpret_cast<const float4*>(&sG[il][4 * tx]);
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(&sA[k][il][4 * ty]);
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][k][c] = __builtin_fmaf(av[r], gv[c], acc[r][k][c]);
            }
        }
    }

    // column factors, weighting and the reduction over this block's 64 columns
    float ysv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ysv[c] = ys[min(j0 + 4 * tx + c, R - 1)];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4 p = sRay[4 * ty + r];
        float m0 = 0.f, mt = 0.f, ms = 0.f, mtt = 0.f, mss = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float s = ysv[c] + p.y;
            const float e = __builtin_amdgcn_exp2f(-((s * s) * p.z));
            const float eu0 = e * acc[r][0][c];
            m0 += eu0;
            ms = __builtin_fmaf(s, eu0, ms);
            mss = __builtin_fmaf(s * s, eu0, mss);
            mt = __builtin_fmaf(e, acc[r][1][c], mt);
            mtt = __builtin_fmaf(e, acc[r][2][c], mtt);
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            m0 += __shfl_xor(m0, d); mt += __shfl_xor(mt, d); ms += __shfl_xor(ms, d);
            mtt += __shfl_xor(mtt, d); mss += __shfl_xor(mss, d);
        }
        const int n = n0 + 4 * ty + r;
        if (tx == 0 && n < N) {
            float* o = moments + (((long)b * JB + jb) * N + n) * HELIO_MOMENT_STRIDE;
            o[0] = m0; o[1] = mt; o[2] = ms; o[3] = mtt; o[4] = mss;
        }
    }
}

int splat_bwd_blocks(int R) { return (R + BW_JT - 1) / BW_JT; }

void launch_splat_bwd(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                      const float* gimg, float* moments, hipStream_t st) {
    const int JB = splat_bwd_blocks(R), NB = (N + BW_NT - 1) / BW_NT;
    hipLaunchKernelGGL(splat_bwd_valu, dim3(JB * NB, B), dim3(256), 0, st, B, N, R, rays, xs, ys, gimg, moments);
}

}  // namespace helio
