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
#include <cstdlib>
#include <type_traits>
#include "helio.h"
#include "step_loss_math.h"
#include "geometry_bwd_ray.h"
#include "cull.h"

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

// ----------------------------------------------------------------------------------------------
// f32 MFMA backward (large problems)
// ----------------------------------------------------------------------------------------------
// The five moments need only TWO contractions of grad_image with a factor table:
//   pass 0:  U[j,n] = Σ_i G[i,j]·A_n[i]   →  (M0, Ms, Mss) = Σ_j E_n[j]·(1, s_j, s_j²)·U[j,n]
//   pass 1:  V[i,n] = Σ_j G[i,j]·E_n[j]   →  (Mt, Mtt)     = Σ_i A_n[i]·(t_i, t_i²)·V[i,n]
// i.e. 2 FMAs per (ray, pixel) instead of the VALU kernel's 3.  Each pass is the forward
// tile kernel with other operands: D[c][n] += Gm[c][k]·F_n[k] on v_mfma_f32_32x32x2_f32,
// c = the image axis that survives (columns in pass 0, rows in pass 1), k = the contracted
// axis, F = the contracted-axis factor.  The surviving axis is put on the MFMA's M
// dimension so that in the C/D register map a lane holds ONE ray (column = lane&31) and 16
// values of c in registers: the epilogue (weight by the other factor, sum over c) is
// in-register apart from one lane^32 exchange.
// Workgroup = 16 waves = 256 c × 256 rays (4×4 waves of 64×64); per 64-deep k-chunk the
// grad-image slab Gm[k][c] and the factor table F[k][ray] are staged in LDS in MFMA operand
// order (lanes ↔ consecutive c / consecutive rays: conflict-free reads and writes).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) const float lds_cf;

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PASS, bool VEC, int WC = 4, int WR = 4>
__device__ __forceinline__ void
splat_bwd_mfma_body(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                    const float* __restrict__ ys, const float* __restrict__ gimg, float* __restrict__ moments,
                    const int* __restrict__ live_counts, const int* __restrict__ live_idx, const int* __restrict__ live_total,
                    const int2* __restrict__ live_map, int live_ct, int tile_rays = CULL_BWD_TILE) {
    static_assert(!(VEC && PASS == 1), "16-byte staging is pass 0's");
    // Row pitch of the two LDS tables.  Pass 1 writes the slab transposed (lanes ↔ k at stride LD):
    // an odd pitch keeps that conflict-free.  Pass 0 writes it along c and stages it 16 bytes at a time
    // (global_load_dwordx4 + ds_write_b128 where R % 4 == 0): a pitch that is a multiple of 4 keeps
    // every row 16-byte aligned; reads (lanes ↔ consecutive c or rays) are conflict-free either way.
    // WC = rows of 64-wide waves along c: 4 → the 256 c × 256 rays tile, 16 waves (images wider than 128);
    // 2 → 128 c × 256 rays, 8 waves, for images of at most 128 pixels across, where half of a 256-wide tile
    // would be padding (R = 128, the reference's default resolution: 0.37 → of the peak with either the padded
    // tile or the small-tile kernel).  The ray extent stays 256: the factor table costs exps per (k, ray),
    // the slab only loads, so the narrow side is the slab's.
    // WR = ray blocks of 64 per workgroup: 4 → 256 rays; 2 → the 128-ray form for the LAST tile of a list when it holds
    // at most 128 rays (8 waves: the slab is staged by half the threads, the factor table and the MFMAs are half) —
    // a list's last tile is half empty on average, and lists are short (≈1100 rays at config 4: cull.h)
    // (the 128-ray form stages 32 k at a time: the same 16 slab values per thread as the 256-ray form, 50 KB of LDS and —
    // held to 128 registers by its launch bounds — two workgroups per CU, each in the other's producer phase)
    // (WR = 1, round 4: 64-ray tiles for fields of 33–128 heliostats on large or many images — 4 waves, 16 k at a time
    // (16 slab values per thread: 32 spilled), 21 KB of LDS, three workgroups per CU by their registers; the slab is
    // staged four times as often per MFMA as in the 256-ray tile)
    constexpr int KC = WR == 1 ? 16 : (WR == 2 ? 32 : 64), T = 64 * WR, TC = 64 * WC, NT = 64 * WC * WR, KPT = KC / WC, NV = KC * TC / NT;
    constexpr int LDG = PASS == 0 ? TC + 4 : TC + 1, LD = PASS == 0 ? T + 4 : T + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // sG[KC][LDG] sF[KC][LD] ccoord[TC]
    float* __restrict__ sCc = smem + KC * (LDG + LD);

    const int c_tiles = (R + TC - 1) / TC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: nothing of it is kept in (or spilled from) VGPRs
    const int lr = lane & 31, lh = lane >> 5;
    // (cull.h) with lists of the rays whose moments are not identically zero, the ray axis of the tiles runs over
    // a list: L entries, entry p is ray lidx[p] (the moment buffer was cleared with the lists: rays not listed read zero).
    // A ray's moments involve no other ray, so which tile computes them changes nothing.  One list per image
    // serving all its c tiles (live_ct = 1), or one per (image, c tile) of THIS pass (live_ct = c_tiles).  The
    // workgroup's (list, ray tile) comes from the table of non-empty tiles, in id order, and the ids past the
    // table leave at once — all at the end of the grid (cull.h, "work order").
    int b = blockIdx.y, bx = blockIdx.x, lst = 0;
    if (live_counts) {
        const unsigned w = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned per = live_ct > 1 ? 1u : (unsigned)c_tiles;
        const unsigned item = w / per;
        if (item >= (unsigned)*live_total) return;
        const int2 e = live_map[item];
        lst = e.x;
        b = live_ct > 1 ? e.x / c_tiles : e.x;
        bx = e.y * c_tiles + (live_ct > 1 ? e.x % c_tiles : (int)(w % per));
    }
    const int c0 = (bx % c_tiles) * TC, n0 = (bx / c_tiles) * tile_rays;     // (tiles are numbered in 256 rays whatever WR — except in the HELIO_BWD_WR2 experiment)
    const int L = live_counts ? live_counts[lst] : N;
    const int* __restrict__ lidx = live_counts ? live_idx + (long)lst * N : nullptr;
    const int wc = (wave / WR) * 64, wn = (wave % WR) * 64;
    const float* __restrict__ ccoord = PASS == 0 ? ys : xs;   // coordinates along c
    const float* __restrict__ kcoord = PASS == 0 ? xs : ys;   // coordinates along k
    const float* __restrict__ G = gimg + (long)b * R * R;

    if (tid < TC) sCc[tid] = ccoord[min(c0 + tid, R - 1)];

    // producer role for the factor table: ray (wave&3)*64 + lane of the tile, KPT = 64 / WC k of every chunk
    const int pr = wn + lane;
    const int pk0 = (wave / WR) * KPT;
    float4 q = make_float4(0.f, 0.f, 1.f, 1e30f);
    if (n0 + pr < L) q = reinterpret_cast<const float4*>(rays)[(long)b * N + (lidx ? lidx[n0 + pr] : n0 + pr)];
    const float sk = __builtin_sqrtf(q.z);
    const float fshift = (PASS == 0 ? q.x : q.y) * sk;
    const float fcc = PASS == 0 ? q.w * q.z : 0.0f;
    lds_f* fdst = (lds_f*)smem + KC * LDG + pk0 * LD + pr;

    // loader role for the grad-image slab Gm[k][c]: 16 dwords per thread and chunk.
    //   pass 0: Gm[k][c] = G[k0+k][c0+c].  R % 4 == 0 (uniform): four 16-byte pieces per thread, piece
    //           p = tid + 1024·v covers k = p>>6, c = 4(p&63)..+3 — a wave reads 1 KB of one image row and
    //           writes 1 KB of one LDS row.  Otherwise dword by dword, lanes ↔ c.
    //   pass 1: Gm[k][c] = G[c0+c][k0+k]:  k = idx&63, c = idx>>6   (lanes ↔ k, stride LD = 257 in LDS)
    float gv[NV];
    auto load_slab = [&](int k0) {
        if constexpr (VEC) {
#pragma unroll
            for (int v = 0; v < NV / 4; ++v) {
                const int p = tid + NT * v;
                const int row = k0 + p / (TC / 4), col = c0 + 4 * (p % (TC / 4));
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < R && col < R) t = *reinterpret_cast<const float4*>(G + (long)row * R + col);
                gv[4 * v] = t.x; gv[4 * v + 1] = t.y; gv[4 * v + 2] = t.z; gv[4 * v + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = tid + NT * v;
                const int k = PASS == 0 ? idx / TC : idx & (KC - 1), c = PASS == 0 ? idx % TC : idx / KC;
                const int row = PASS == 0 ? k0 + k : c0 + c, col = PASS == 0 ? c0 + c : k0 + k;
                gv[v] = (row < R && col < R) ? G[(long)row * R + col] : 0.0f;
            }
        }
    };
    auto store_slab = [&]() {
        if constexpr (VEC) {
#pragma unroll
            for (int v = 0; v < NV / 4; ++v) {
                const int p = tid + NT * v;
                *reinterpret_cast<float4*>(smem + (p / (TC / 4)) * LDG + 4 * (p % (TC / 4))) =
                    make_float4(gv[4 * v], gv[4 * v + 1], gv[4 * v + 2], gv[4 * v + 3]);
            }
        } else {
            lds_f* dst = (lds_f*)smem;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = tid + NT * v;
                const int k = PASS == 0 ? idx / TC : idx & (KC - 1), c = PASS == 0 ? idx % TC : idx / KC;
                dst[k * LDG + c] = gv[v];
            }
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;

    // consumer: A operand = Gm[k = 2kp + lh][c = wc + 32·blk + lr], B operand = F[k][ray = wn + 32·blk + lr]
    lds_cf* pg = (lds_cf*)smem + lh * LDG + wc + lr;
    lds_cf* pf = (lds_cf*)smem + KC * LDG + lh * LD + wn + lr;
    lds_cf* pg1 = pg + 32;
    lds_cf* pf1 = pf + 32;
    asm volatile("" : "+v"(pg));
    asm volatile("" : "+v"(pf));
    asm volatile("" : "+v"(pg1));
    asm volatile("" : "+v"(pf1));

    // (a PERSISTENT form — one workgroup per CU walking its 32 tiles, no workgroup turnover — was built and
    // measured at 8.70 ms against 8.33: the loop-carried tile state does not fit the 128-register budget
    // of a 16-wave workgroup — 21 spilled VGPRs and 25 SGPRs in pass 1; not kept.)
    // (skipping the MFMAs of ray blocks past the last heliostat — N = 2000 fills 7.8 of its 8 tiles — was
    // tried as a wave-uniform choice between a 4- and a 2-MFMA loop: the second unrolled loop cost 72
    // spilled registers; not kept)
    // the 16 contraction coordinates a wave needs per chunk are wave-uniform: scalar loads, no
    // LDS staging (and one barrier fewer per chunk)
    load_slab(0);
    for (int k0 = 0; k0 < R; k0 += KC) {
        float kc[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) kc[j] = kcoord[min(k0 + pk0 + j, R - 1)];
        __syncthreads();                                   // previous chunk consumed
        store_slab();
        if (k0 + KC < R) load_slab(k0 + KC);               // in flight during the chunk
        if constexpr (!VEC) {       // (the packed form below costs these two instantiations a spilled register)
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const float t = __builtin_fmaf(kc[j], sk, fshift);
                float f = __builtin_amdgcn_exp2f(-__builtin_fmaf(t, t, fcc));
                if (k0 + pk0 + j >= R) f = 0.0f;           // rows/cols past the image contract nothing
                fdst[j * LD] = f;
            }
        } else {
            // factor pairs: the two fused multiply-adds of two factors are one v_pk_fma_f32 each (the same
            // IEEE fma per component; producer instructions are MFMA time on this chip)
            const f32x2 sk2 = {sk, sk}, sh2 = {fshift, fshift}, cc2 = {fcc, fcc};
#pragma unroll
            for (int j = 0; j < KPT; j += 2) {
                const f32x2 kc2 = {kc[j], kc[j + 1]};
                const f32x2 t = __builtin_elementwise_fma(kc2, sk2, sh2);
                const f32x2 a = __builtin_elementwise_fma(t, t, cc2);
                float f0 = __builtin_amdgcn_exp2f(-a.x), f1 = __builtin_amdgcn_exp2f(-a.y);
                if (k0 + pk0 + j >= R) f0 = 0.0f;          // rows/cols past the image contract nothing
                if (k0 + pk0 + j + 1 >= R) f1 = 0.0f;
                fdst[j * LD] = f0;
                fdst[(j + 1) * LD] = f1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < KC / 2; ++kp) {
            const float g0 = pg[kp * 2 * LDG], g1 = pg1[kp * 2 * LDG];
            const float f0 = pf[kp * 2 * LD], f1 = pf1[kp * 2 * LD];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, f0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, f1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, f0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, f1, acc[3], 0, 0, 0);
        }
    }

    // epilogue: this lane's ray is column lr of ray block nb; the 16 registers of an accumulator block
    // are c = (e&3) + 8(e>>2) + 4·lh of c block cb.  Two values of c at a time (registers e, e+1 are
    // neighbours in c): the weights, the products and the three running sums are formed on float pairs
    // (v_pk_add/mul/fma_f32: one instruction per pair; no MFMA runs beside them here).
    const int JB = (R + 63) / 64;
    const int cblock = (c0 + wc) / 64;
    if (c0 + wc >= R) return;                              // (wave-uniform) nothing of the image here
    // the lane's place in the wave, read again: keeping lr / lh alive across the loop cost two spilled registers
    const int lane2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int lr2 = lane2 & 31, lh2 = lane2 >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int p = n0 + wn + 32 * nb + lr2;
        const int n = p < L ? (lidx ? lidx[p] : p) : N;
        float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) h = reinterpret_cast<const float4*>(rays)[(long)b * N + n];
        const float hshift = PASS == 0 ? h.y : h.x;
        const float hcc = PASS == 0 ? 0.0f : h.w;
        const f32x2 sh = {hshift, hshift}, cc2 = {hcc, hcc}, nk = {-h.z, -h.z};
        f32x2 m0 = {0.f, 0.f}, m1 = {0.f, 0.f}, m2 = {0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const int cl = wc + 32 * cb + (e & 3) + 8 * (e >> 2) + 4 * lh2;
                const f32x2 s = *reinterpret_cast<const f32x2*>(&sCc[cl]) + sh;
                const f32x2 ss = s * s;
                const f32x2 arg = (ss + cc2) * nk;
                const f32x2 a = {acc[2 * cb + nb][e], acc[2 * cb + nb][e + 1]};
                const f32x2 ex = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                const f32x2 w = ex * a;
                m0 += w;
                m1 += s * w;
                m2 += ss * w;
            }
        }
        float t0 = m0.x + m0.y, t1 = m1.x + m1.y, t2 = m2.x + m2.y;
        t0 += __shfl_xor(t0, 32); t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
        if (lh2 == 0 && n < N) {
            float* o = moments + (((long)b * JB + cblock) * N + n) * HELIO_MOMENT_STRIDE;
            if (PASS == 0) { o[0] = t0; o[2] = t1; o[4] = t2; }
            else { o[1] = t1; o[3] = t2; }
        }
    }
}

// ---- the same tile with the two LDS tables DOUBLE-BUFFERED (round 4; opt-in, HELIO_BWD_DB=1: it did NOT pay) -------
// The body above alternates two phases per 64-deep chunk, separated by barriers: every wave stores its share of the slab
// and computes its share of the factor table (16 exponentials and their fused multiply-adds per thread: VALU), then
// every wave issues its 128 MFMAs.  In the first phase the matrix pipe of all four SIMDs idles — PMC (round 2): pipe busy
// 85.6 % of the kernel's cycles.  Here a chunk is 32 deep and the tables exist twice (2 × 66.5 KB): while a wave issues
// the MFMAs of chunk c out of one pair of tables it produces chunk c + 1 into the other — one factor after every eight
// MFMAs, in the shadow of the pipe — and ONE barrier per chunk hands the pairs over.  Same operands, same order of the
// contracted axis, same epilogue: the moments are the bits of the single-buffered body.
// MEASURED (config 4, one box, profiles/r04_e_bwd_db.txt): dense 8.29 ms single-buffered; this body 9.15 ms in its first
// form (every pair of operand reads waited for in front of its four MFMAs), 8.71 ms with the operands of k-pair kp + 1
// requested before the MFMAs of kp and the epilogue's rays fetched at kernel start (114–118 VGPRs, no spill) — 5 % BEHIND:
// the producer's vector instructions are not free beside this chip's 16-pass f32 MFMAs (they take the SIMD's issue slots
// whichever wave they come from), so interleaving them moves the producer phase's cycles into the MFMA phase instead of
// hiding them, and the per-chunk pointer arithmetic and branches come on top.  The single-buffered body stays the default.
template <int PASS, bool VEC, int WC>
__device__ __forceinline__ void
splat_bwd_mfma_body_db(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                       const float* __restrict__ ys, const float* __restrict__ gimg, float* __restrict__ moments,
                       const int* __restrict__ live_counts, const int* __restrict__ live_idx, const int* __restrict__ live_total,
                       const int2* __restrict__ live_map, int live_ct) {
    static_assert(!(VEC && PASS == 1), "16-byte staging is pass 0's");
    constexpr int WR = 4, KC = 32, T = 64 * WR, TC = 64 * WC, NT = 64 * WC * WR, KPT = KC / WC, NV = KC * TC / NT;
    static_assert(NV == 8 && (KPT == 8 || KPT == 16), "8 slab values and 8 or 16 factors per thread and chunk");
    constexpr int LDG = PASS == 0 ? TC + 4 : TC + 1, LD = PASS == 0 ? T + 4 : T + 1;
    constexpr int BUF = KC * (LDG + LD);                            // floats of one pair of tables
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 × { sG[KC][LDG] sF[KC][LD] }  ccoord[TC]
    float* __restrict__ sCc = smem + 2 * BUF;

    const int c_tiles = (R + TC - 1) / TC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    int b = blockIdx.y, bx = blockIdx.x, lst = 0;
    if (live_counts) {                                              // (cull.h: the tile comes from the work map)
        const unsigned w = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned per = live_ct > 1 ? 1u : (unsigned)c_tiles;
        const unsigned item = w / per;
        if (item >= (unsigned)*live_total) return;
        const int2 e = live_map[item];
        lst = e.x;
        b = live_ct > 1 ? e.x / c_tiles : e.x;
        bx = e.y * c_tiles + (live_ct > 1 ? e.x % c_tiles : (int)(w % per));
    }
    const int c0 = (bx % c_tiles) * TC, n0 = (bx / c_tiles) * CULL_BWD_TILE;
    const int L = live_counts ? live_counts[lst] : N;
    const int* __restrict__ lidx = live_counts ? live_idx + (long)lst * N : nullptr;
    const int wc = (wave / WR) * 64, wn = (wave % WR) * 64;
    const float* __restrict__ ccoord = PASS == 0 ? ys : xs;
    const float* __restrict__ kcoord = PASS == 0 ? xs : ys;
    const float* __restrict__ G = gimg + (long)b * R * R;

    if (tid < TC) sCc[tid] = ccoord[min(c0 + tid, R - 1)];

    // producer role for the factor table: ray wn + lane of the tile, KPT k of every chunk
    const int pr = wn + lane;
    const int pk0 = (wave / WR) * KPT;
    float4 q = make_float4(0.f, 0.f, 1.f, 1e30f);
    if (n0 + pr < L) q = reinterpret_cast<const float4*>(rays)[(long)b * N + (lidx ? lidx[n0 + pr] : n0 + pr)];
    const float sk = __builtin_sqrtf(q.z);
    const float fshift = (PASS == 0 ? q.x : q.y) * sk;
    const float fcc = PASS == 0 ? q.w * q.z : 0.0f;
    const int foff = KC * LDG + pk0 * LD + pr;                      // this thread's first factor, inside a pair of tables
    // the two rays this lane's EPILOGUE belongs to (column lr of ray blocks 0 and 1 of the wave), requested now: fetched
    // after the loop they are two dependent misses (list entry, then ray) in front of every workgroup's last microseconds
    int en[2];
    float4 eh[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int p = n0 + wn + 32 * nb + lr;
        en[nb] = p < L ? (lidx ? lidx[p] : p) : N;
        eh[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (en[nb] < N) eh[nb] = reinterpret_cast<const float4*>(rays)[(long)b * N + en[nb]];
    }

    // loader role for the grad-image slab Gm[k][c]: 8 dwords per thread and chunk (layouts as in the body above)
    float gv[NV];
    auto load_slab = [&](int k0) {
        if constexpr (VEC) {
#pragma unroll
            for (int v = 0; v < NV / 4; ++v) {
                const int p = tid + NT * v;
                const int row = k0 + p / (TC / 4), col = c0 + 4 * (p % (TC / 4));
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < R && col < R) t = *reinterpret_cast<const float4*>(G + (long)row * R + col);
                gv[4 * v] = t.x; gv[4 * v + 1] = t.y; gv[4 * v + 2] = t.z; gv[4 * v + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = tid + NT * v;
                const int k = PASS == 0 ? idx / TC : idx & (KC - 1), c = PASS == 0 ? idx % TC : idx / KC;
                const int row = PASS == 0 ? k0 + k : c0 + c, col = PASS == 0 ? c0 + c : k0 + k;
                gv[v] = (row < R && col < R) ? G[(long)row * R + col] : 0.0f;
            }
        }
    };
    auto store_slab = [&](float* __restrict__ buf) {
        if constexpr (VEC) {
#pragma unroll
            for (int v = 0; v < NV / 4; ++v) {
                const int p = tid + NT * v;
                *reinterpret_cast<float4*>(buf + (p / (TC / 4)) * LDG + 4 * (p % (TC / 4))) =
                    make_float4(gv[4 * v], gv[4 * v + 1], gv[4 * v + 2], gv[4 * v + 3]);
            }
        } else {
            lds_f* dst = (lds_f*)buf;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = tid + NT * v;
                const int k = PASS == 0 ? idx / TC : idx & (KC - 1), c = PASS == 0 ? idx % TC : idx / KC;
                dst[k * LDG + c] = gv[v];
            }
        }
    };
    // factor j of the chunk at k0 → the pair of tables at `buf`; kc = the chunk's coordinate pk0 + j (wave-uniform)
    auto factor = [&](float* __restrict__ buf, int k0, int j, float kc) {
        const float t = __builtin_fmaf(kc, sk, fshift);
        float f = __builtin_amdgcn_exp2f(-__builtin_fmaf(t, t, fcc));
        if (k0 + pk0 + j >= R) f = 0.0f;                            // rows / columns past the image contract nothing
        ((lds_f*)buf)[foff + j * LD] = f;
    };

    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;

    // chunk 0 into pair 0; chunk 1's slab into the registers, its coordinates into scalars
    float kc[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) kc[j] = kcoord[min(pk0 + j, R - 1)];
    load_slab(0);
    store_slab(smem);
#pragma unroll
    for (int j = 0; j < KPT; ++j) factor(smem, 0, j, kc[j]);
    load_slab(KC);
#pragma unroll
    for (int j = 0; j < KPT; ++j) kc[j] = kcoord[min(KC + pk0 + j, R - 1)];
    __syncthreads();

    const int goff = lh * LDG + wc + lr, foff_c = KC * LDG + lh * LD + wn + lr;      // consumer operands inside a pair
    // One chunk: 16 k-pairs of 4 MFMAs out of the pair of tables `cur`; the operands of k-pair kp + 1 are requested
    // BEFORE the MFMAs of k-pair kp are issued (the LDS round trip then runs beside them: the first form of this loop
    // waited for every pair of reads in front of its four MFMAs and lost 10 % to it), and with MORE the next chunk is
    // produced into the other pair of tables in eight pieces, one behind every second k-pair's MFMAs.
    auto chunk = [&](auto more_c, int k0) {
        constexpr bool MORE = decltype(more_c)::value;
        const int cur = (k0 / KC) & 1;
        float* __restrict__ bufn = smem + (cur ^ 1) * BUF;
        lds_cf* pg = (lds_cf*)smem + cur * BUF + goff;
        lds_cf* pf = (lds_cf*)smem + cur * BUF + foff_c;
        float g0 = pg[0], g1 = pg[32], f0 = pf[0], f1 = pf[32];
#pragma unroll
        for (int kp = 0; kp < KC / 2; ++kp) {
            float g0n = 0.f, g1n = 0.f, f0n = 0.f, f1n = 0.f;
            if (kp + 1 < KC / 2) {
                g0n = pg[(kp + 1) * 2 * LDG]; g1n = pg[(kp + 1) * 2 * LDG + 32];
                f0n = pf[(kp + 1) * 2 * LD]; f1n = pf[(kp + 1) * 2 * LD + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, f0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, f1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, f0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, f1, acc[3], 0, 0, 0);
            if constexpr (MORE) {
                if ((kp & 1) == 1) {
                    const int seg = kp >> 1;
                    if (seg == 0) store_slab(bufn);                 // chunk c + 1's slab (fetched a chunk ago)
                    if (seg == 1) load_slab(k0 + 2 * KC);           // chunk c + 2's: in flight for a whole chunk
#pragma unroll
                    for (int jj = 0; jj < KPT / 8; ++jj) {
                        const int j = seg * (KPT / 8) + jj;
                        factor(bufn, k0 + KC, j, kc[j]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            g0 = g0n; g1 = g1n; f0 = f0n; f1 = f1n;
        }
        if constexpr (MORE) {
#pragma unroll
            for (int j = 0; j < KPT; ++j) kc[j] = kcoord[min(k0 + 2 * KC + pk0 + j, R - 1)];
        }
        __syncthreads();                                            // pair cur consumed by everybody, pair cur^1 complete
    };
    int k0 = 0;
    for (; k0 + KC < R; k0 += KC) chunk(std::true_type{}, k0);
    chunk(std::false_type{}, k0);

    // epilogue: as in the body above
    const int JB = (R + 63) / 64;
    const int cblock = (c0 + wc) / 64;
    if (c0 + wc >= R) return;
    const int lane2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int lh2 = lane2 >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = en[nb];
        const float4 h = eh[nb];
        const float hshift = PASS == 0 ? h.y : h.x;
        const float hcc = PASS == 0 ? 0.0f : h.w;
        const f32x2 sh = {hshift, hshift}, cc2 = {hcc, hcc}, nk = {-h.z, -h.z};
        f32x2 m0 = {0.f, 0.f}, m1 = {0.f, 0.f}, m2 = {0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const int cl = wc + 32 * cb + (e & 3) + 8 * (e >> 2) + 4 * lh2;
                const f32x2 s = *reinterpret_cast<const f32x2*>(&sCc[cl]) + sh;
                const f32x2 ss = s * s;
                const f32x2 arg = (ss + cc2) * nk;
                const f32x2 a = {acc[2 * cb + nb][e], acc[2 * cb + nb][e + 1]};
                const f32x2 ex = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                const f32x2 w = ex * a;
                m0 += w;
                m1 += s * w;
                m2 += ss * w;
            }
        }
        float t0 = m0.x + m0.y, t1 = m1.x + m1.y, t2 = m2.x + m2.y;
        t0 += __shfl_xor(t0, 32); t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
        if (lh2 == 0 && n < N) {
            float* o = moments + (((long)b * JB + cblock) * N + n) * HELIO_MOMENT_STRIDE;
            if (PASS == 0) { o[0] = t0; o[2] = t1; o[4] = t2; }
            else { o[1] = t1; o[3] = t2; }
        }
    }
}

template <int PASS, bool VEC, int WC = 4>
__global__ void __launch_bounds__(256 * WC)
splat_bwd_mfma(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
               const float* __restrict__ ys, const float* __restrict__ gimg, float* __restrict__ moments,
               const int* __restrict__ live_counts, const int* __restrict__ live_idx, const int* __restrict__ live_total,
               const int2* __restrict__ live_map, int live_ct) {
    splat_bwd_mfma_body<PASS, VEC, WC>(B, N, R, rays, xs, ys, gimg, moments, live_counts, live_idx, live_total, live_map, live_ct);
}

// Both passes in ONE launch (blockIdx.z = pass; the grid is walked x, y, z: all of pass 0, then pass 1): the tail
// of pass 0 — the last, partly filled round of its workgroups — runs beside the head of pass 1.  set_lists /
// map_stride: where pass 1's lists and its part of the work map start when every (pass, c tile) has lists of its
// own (cull.h), else unused.  WR = 2: the 128-ray form, run over the map of short last tiles.
// DB: the double-buffered body (256-ray tiles only)
template <bool VEC, int WC, int WR, bool DB = false>
// (launch bounds — 256-wide tiles: four waves a SIMD, one 16-wave workgroup or two 8-wave ones per CU; 128-wide: two, one
// 8-wave workgroup with its 101 KB of LDS)
__global__ void __launch_bounds__(64 * WC * WR, WR == 1 ? 3 : (WC == 4 ? 4 : 2))
splat_bwd_mfma_both(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                    const float* __restrict__ ys, const float* __restrict__ gimg, float* __restrict__ moments,
                    const int* __restrict__ live_counts, const int* __restrict__ live_idx, const int* __restrict__ live_total,
                    const int2* __restrict__ live_map, int live_ct, long set_lists, long map_stride, int tile_rays) {
    static_assert(!DB || WR == 4, "the double-buffered body is the 256-ray tile's");
    if (blockIdx.z == 0) {
        if constexpr (DB) splat_bwd_mfma_body_db<0, VEC, WC>(B, N, R, rays, xs, ys, gimg, moments, live_counts, live_idx, live_total, live_map, live_ct);
        else splat_bwd_mfma_body<0, VEC, WC, WR>(B, N, R, rays, xs, ys, gimg, moments, live_counts, live_idx, live_total, live_map, live_ct, tile_rays);
    } else {
        const bool own = live_counts && live_ct > 1;
        const int* lc = own ? live_counts + set_lists : live_counts;
        const int* li = own ? live_idx + set_lists * N : live_idx;
        const int* lt = own ? live_total + 1 : live_total;
        const int2* lm = own ? live_map + map_stride : live_map;
        if constexpr (DB) splat_bwd_mfma_body_db<1, false, WC>(B, N, R, rays, xs, ys, gimg, moments, lc, li, lt, lm, live_ct);
        else splat_bwd_mfma_body<1, false, WC, WR>(B, N, R, rays, xs, ys, gimg, moments, lc, li, lt, lm, live_ct, tile_rays);
    }
}

// Small problems (config 3: B=25, N=50, R=128) are bound by the latency of ONE wave, not by the matrix
// pipe: the round-1 form of this kernel (64 c × 64 rays per 4-wave workgroup, a 128-deep k-chunk staged
// in LDS, every wave 64 dependent MFMAs) took 14.5–16 µs on 100 of the 256 CUs.  Same two contractions,
// one launch (blockIdx.z = pass), reshaped like the fused forward kernel:
//   * a workgroup is 64 c × 32 rays (one partial block of the moment buffer, one ray block) and KS waves
//     that SPLIT THE CONTRACTED AXIS between them: config 3 runs 200 workgroups of 4 waves, 32 MFMAs each;
//   * no LDS staging and no barrier in front of the MFMAs: the grad-image operand goes from global
//     memory straight into the MFMA's A register (lane ↔ c: coalesced rows in pass 0; in pass 1 a lane
//     reads its own image row, 16 bytes at a time where R % 4 == 0), the factor operand is computed in
//     registers by the lane that owns (k, ray), and all loads of a group of 8 k-pairs are issued ahead of
//     its MFMAs;
//   * every wave weights its partial block with the other-axis factor and reduces it over c in
//     registers, so that only three numbers per ray cross the workgroup (LDS, one barrier), summed in
//     fixed wave order.
// One wave's share of one contraction: the 64 c × 32 rays block at (c0, n0) of image `G`, contracted over
// k in [k_begin, k_end) (even bounds), weighted by the other-axis factor and reduced over the block's 64 c
// in registers → (m0, m1, m2) for the ray n0 + (lane & 31), complete in BOTH halves of the wave.
// `sCc_wave`: 64 floats of LDS private to the wave.
// NRB ray blocks of 32 share the wave's grad-image operands (one load of the 64 c × k slab feeds 2·NRB MFMAs
// per k-pair): NRB = 1 where latency is everything, 2 or 4 where there are ray blocks enough to keep the
// chip's workgroup slots full anyway — then the slab is fetched from L2 half / a quarter as often.
// → m[rb][0..2] for the ray n0 + 32·rb + (lane & 31).
// LISTS (cull.h): the ray axis runs over the image's list of rays whose footprint is not identically zero — position
// p < L is ray lidx[p] — and nn[rb] returns the ray this lane's results belong to (-1: none).  A template
// parameter, not a null pointer: with the choice made at run time the DENSE kernel lost 20–26 % where one wave's
// latency is the whole time (B = 4, N = 1000, R = 512: 68 → 87 µs; the ray request behind a select).
template <int PASS, int NRB, bool LISTS = false>
__device__ __forceinline__ void small_wave_partial_n(int N, int R, const float* __restrict__ rays_b,
                                                     const float* __restrict__ xs, const float* __restrict__ ys,
                                                     const float* __restrict__ G, int c0, int n0, int k_begin, int k_end,
                                                     float* __restrict__ sCc_wave, float (&m)[NRB][3],
                                                     const int* __restrict__ lidx = nullptr, int L = 0, int* nn = nullptr) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const float* __restrict__ ccoord = PASS == 0 ? ys : xs;   // coordinates along c (the axis that survives)
    const float* __restrict__ kcoord = PASS == 0 ? xs : ys;   // coordinates along k (the contracted axis)
    // this lane's ray (B operand and epilogue) and its c coordinate: requested first, used only after the
    // first group of grad-image loads has been issued — every first touch of memory in a freshly launched
    // kernel costs ≈900 cycles, so none of them may wait for another
    float4 qraw[NRB];
    const int lim = LISTS ? L : N;                            // positions from here on are padding: they contribute nothing
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        if constexpr (LISTS) {
            const int p = n0 + 32 * rb + lr;
            const int n = lidx[min(p, max(L - 1, 0))];
            qraw[rb] = reinterpret_cast<const float4*>(rays_b)[n];
            nn[rb] = p >= L ? -1 : n;
        } else {
            qraw[rb] = reinterpret_cast<const float4*>(rays_b)[min(n0 + 32 * rb + lr, N - 1)];
        }
    }
    const float ccv = ccoord[min(c0 + lane, R - 1)];

    f32x16 acc0[NRB], acc1[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[rb][e] = 0.0f; acc1[rb][e] = 0.0f; }

    const int ca = c0 + lr, cb = c0 + 32 + lr;                  // this lane's two c (A operands of the two blocks)
    const bool vec = PASS == 1 && (R & 3) == 0;                 // (uniform) pass 1 may read 16-byte row segments
    for (int k0 = k_begin; k0 < k_end; k0 += 16) {              // groups of 8 k-pairs
        float ga[8], gb[8], kc[8];
        // every load is unconditional, from a clamped (valid) address; what lies outside the image or
        // this wave's part is zeroed AFTER the whole group has been fetched (hipcc turns "in range ? load
        // : 0" into a branch around each load, and then waits for each on its own)
        const long rowa = (long)min(ca, R - 1) * R, rowb = (long)min(cb, R - 1) * R;
        if (vec) {
            // G[c][k0 .. k0+15]: lane (c, lh) uses elements lh and 2 + lh of every 16-byte segment
            float4 va[4], vb[4];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int kk = min(k0 + 4 * j4, R - 4);       // (R % 4 == 0: a whole segment is in or out)
                va[j4] = *reinterpret_cast<const float4*>(G + rowa + kk);
                vb[j4] = *reinterpret_cast<const float4*>(G + rowb + kk);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) kc[j] = kcoord[min(k0 + 2 * j + lh, R - 1)];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
                asm volatile("" :: "v"(va[j4].x), "v"(va[j4].y), "v"(va[j4].z), "v"(va[j4].w), "v"(vb[j4].x), "v"(vb[j4].y),
                             "v"(vb[j4].z), "v"(vb[j4].w), "v"(kc[2 * j4]), "v"(kc[2 * j4 + 1]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const bool ok = k0 + 4 * j4 < k_end;
                ga[2 * j4] = (ok && ca < R) ? (lh ? va[j4].y : va[j4].x) : 0.0f;
                ga[2 * j4 + 1] = (ok && ca < R) ? (lh ? va[j4].w : va[j4].z) : 0.0f;
                gb[2 * j4] = (ok && cb < R) ? (lh ? vb[j4].y : vb[j4].x) : 0.0f;
                gb[2 * j4 + 1] = (ok && cb < R) ? (lh ? vb[j4].w : vb[j4].z) : 0.0f;
            }
        } else {
            float ra[8], rb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = min(k0 + 2 * j + lh, R - 1);
                ra[j] = PASS == 0 ? G[(long)kk * R + min(ca, R - 1)] : G[rowa + kk];
                rb[j] = PASS == 0 ? G[(long)kk * R + min(cb, R - 1)] : G[rowb + kk];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) kc[j] = kcoord[min(k0 + 2 * j + lh, R - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(ra[j]), "v"(rb[j]), "v"(kc[j]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = k0 + 2 * j + lh < k_end;
                ga[j] = (ok && ca < R) ? ra[j] : 0.0f;
                gb[j] = (ok && cb < R) ? rb[j] : 0.0f;
            }
        }
        // (the ray's constants are derived HERE, behind the group's loads: computed in front of the loop
        // they would put the ray's own load latency in front of every other load)
        float f[NRB][8];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            float4 q = qraw[rb];
            asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));
            if (n0 + 32 * rb + lr >= lim) q = make_float4(0.f, 0.f, 1.f, 1e30f);      // padding ray: factor = exp2(-1e30) = 0
            const float sk = __builtin_sqrtf(q.z);
            const float fshift = (PASS == 0 ? q.x : q.y) * sk;
            const float fcc = PASS == 0 ? q.w * q.z : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = __builtin_fmaf(kc[j], sk, fshift);
                f[rb][j] = (k0 + 2 * j + lh < k_end) ? __builtin_amdgcn_exp2f(-__builtin_fmaf(t, t, fcc)) : 0.0f;
            }
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(f[rb][j]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (k0 + 2 * j >= k_end) break;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                acc0[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j], f[rb][j], acc0[rb], 0, 0, 0);
                acc1[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(gb[j], f[rb][j], acc1[rb], 0, 0, 0);
            }
        }
    }

    // epilogue: lane = ray (column lr), registers = 16 values of c per block; weight by the other-axis
    // factor and reduce over this wave's 64 c
    sCc_wave[lane] = ccv;                                // wave-private: LDS is in order within a wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        const float hshift = PASS == 0 ? qraw[rb].y : qraw[rb].x;
        const float hcc = PASS == 0 ? 0.0f : qraw[rb].w;
        const float hk = n0 + 32 * rb + lr < lim ? qraw[rb].z : 0.0f;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cl = 32 * blk + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float s = sCc_wave[cl] + hshift;
                const float w = __builtin_amdgcn_exp2f(-(__builtin_fmaf(s, s, hcc) * hk)) * (blk == 0 ? acc0[rb][e] : acc1[rb][e]);
                m0 += w;
                m1 = __builtin_fmaf(s, w, m1);
                m2 = __builtin_fmaf(s * s, w, m2);
            }
        m[rb][0] = m0 + __shfl_xor(m0, 32); m[rb][1] = m1 + __shfl_xor(m1, 32); m[rb][2] = m2 + __shfl_xor(m2, 32);
    }
}

// (one ray block: what the latency-bound callers use)
template <int PASS>
__device__ __forceinline__ void small_wave_partial(int N, int R, const float* __restrict__ rays_b,
                                                   const float* __restrict__ xs, const float* __restrict__ ys,
                                                   const float* __restrict__ G, int c0, int n0, int k_begin, int k_end,
                                                   float* __restrict__ sCc_wave, float& m0, float& m1, float& m2) {
    float m[1][3];
    small_wave_partial_n<PASS, 1>(N, R, rays_b, xs, ys, G, c0, n0, k_begin, k_end, sCc_wave, m);
    m0 = m[0][0]; m1 = m[0][1]; m2 = m[0][2];
}

// A workgroup = KS × RBW waves: KS of them split the contracted axis of one (64 c, 32·NRB rays) block — the
// latency-bound end: config 3 runs KS = 4, RBW = 1 —, RBW such groups sit side by side along the rays.  KS = 1
// (no split, no LDS reduce: RBW independent waves) is the throughput end: every wave pays the epilogue — 32
// exps and ≈400 vector instructions per ray block, PMC: more SIMD time than its MFMAs at KS = 4 — once per
// R MFMAs instead of once per R / 4.
template <int PASS, int KS, int NRB, int RBW, bool LISTS>
__device__ __forceinline__ void splat_bwd_small_body(int N, int R, const float* __restrict__ rays,
                                                     const float* __restrict__ xs, const float* __restrict__ ys,
                                                     const float* __restrict__ gimg, float* __restrict__ moments,
                                                     float* smem, const int* __restrict__ live_counts,
                                                     const int* __restrict__ live_idx) {
    constexpr int NW = KS * RBW, RPG = 32 * NRB;      // waves per workgroup; rays per k-split group
    float* __restrict__ sCc = smem;                // [NW][64] c coordinates, one private copy per wave
    float* __restrict__ sRed = smem + NW * 64;     // [RBW][KS][RPG rays][3]

    const int JB = (R + 63) / 64;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kpart = wave % KS, grp = wave / KS;
    const int lr = lane & 31, lh = lane >> 5;
    const int c0 = (blockIdx.x % JB) * 64, n0 = ((blockIdx.x / JB) * RBW + grp) * RPG;
    const float* __restrict__ G = gimg + (long)b * R * R;
    // (cull.h) per-image list of the rays that are not identically zero: workgroups past its end leave at once
    const int L = LISTS ? live_counts[b] : N;
    const int* __restrict__ lidx = LISTS ? live_idx + (long)b * N : nullptr;
    if constexpr (LISTS)
        if ((int)(blockIdx.x / JB) * RBW * RPG >= L) return;

    // this wave's part of the contracted axis: k-pairs dealt evenly, in multiples of 2 (so that a wave
    // starts at a multiple of 4: 16-byte row segments in pass 1)
    const int pairs = (R + 1) >> 1;
    const int per = (((pairs + KS - 1) / KS) + 1) & ~1;
    const int k_begin = min(R, 2 * per * kpart), k_end = min(R, 2 * per * (kpart + 1));

    float m[NRB][3];
    int nn[NRB];
    small_wave_partial_n<PASS, NRB, LISTS>(N, R, rays + 4l * b * N, xs, ys, G, c0, n0, k_begin, k_end, sCc + wave * 64, m, lidx, L, nn);
    if constexpr (KS == 1) {
        if (lh == 0) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                int n = n0 + 32 * rb + lr;
                if constexpr (LISTS) n = nn[rb];
                else if (n >= N) n = -1;
                if (n >= 0) {
                    float* o = moments + (((long)b * JB + c0 / 64) * N + n) * HELIO_MOMENT_STRIDE;
                    if (PASS == 0) { o[0] = m[rb][0]; o[2] = m[rb][1]; o[4] = m[rb][2]; }
                    else { o[1] = m[rb][1]; o[3] = m[rb][2]; }
                }
            }
        }
        return;
    }
    if (lh == 0) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            float* r = sRed + ((grp * KS + kpart) * RPG + 32 * rb + lr) * 3;
            r[0] = m[rb][0]; r[1] = m[rb][1]; r[2] = m[rb][2];
        }
    }
    __syncthreads();
    if (tid < RBW * RPG) {                                          // fixed order over the KS k-parts
        const int g = tid / RPG, rl = tid % RPG;
        const int p = ((blockIdx.x / JB) * RBW + g) * RPG + rl;
        const int n = p < L ? (LISTS ? lidx[p] : p) : N;
        if (n < N) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < KS; ++w) {
                const float* r = sRed + ((g * KS + w) * RPG + rl) * 3;
                if (w == 0) { t0 = r[0]; t1 = r[1]; t2 = r[2]; } else { t0 += r[0]; t1 += r[1]; t2 += r[2]; }
            }
            float* o = moments + (((long)b * JB + c0 / 64) * N + n) * HELIO_MOMENT_STRIDE;
            if (PASS == 0) { o[0] = t0; o[2] = t1; o[4] = t2; }
            else { o[1] = t1; o[3] = t2; }
        }
    }
}

template <int KS, int NRB = 1, int RBW = 1, bool LISTS = false>
__global__ void __launch_bounds__(64 * KS * RBW)
splat_bwd_mfma_small(int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                     const float* __restrict__ ys, const float* __restrict__ gimg, float* __restrict__ moments,
                     const int* __restrict__ live_counts, const int* __restrict__ live_idx) {
    __shared__ float smem[KS * RBW * 64 + KS * RBW * 32 * NRB * 3];
    if (blockIdx.z == 0) splat_bwd_small_body<0, KS, NRB, RBW, LISTS>(N, R, rays, xs, ys, gimg, moments, smem, live_counts, live_idx);
    else splat_bwd_small_body<1, KS, NRB, RBW, LISTS>(N, R, rays, xs, ys, gimg, moments, smem, live_counts, live_idx);
}

// ----------------------------------------------------------------------------------------------
// Small problems, the WHOLE backward of the render in one launch: moments + geometry adjoint.
// ----------------------------------------------------------------------------------------------
// At config 3 the backward was two launches of latency — splat_bwd_mfma_small (6.6 µs: first touch of the
// rays and of the grad image, 32 MFMAs a wave, epilogue, LDS reduce, store of the partial moments) and
// geometry_bwd_kernel (4.2 µs: first touch again — of the moments just written —, the trace and its
// adjoint), with a 1.5 µs kernel boundary between them.  Here a workgroup owns 32 rays of one sun
// COMPLETELY: its 16 waves are the (pass, 64-wide c block, k part) combinations the small kernel spreads
// over workgroups — CT c blocks (R <= 64·CT), KS = 8 / CT k parts, every wave exactly small_wave_partial —
// so all five moments of its rays meet in LDS, and lanes 0..31 of wave 0 run the geometry adjoint on them
// at once.  Those lanes request their ray's inputs before the contraction starts; the moments never
// touch memory.  Sums in fixed order (k parts, then c blocks — the order the two-launch path adds them in).
bool splat_bwd_is_few(int B, int N);
bool splat_bwd_is_few(int B, int N, int R);

// Where the LDS-tile kernel runs in 64-RAY tiles (variant 12, round 4: splat_bwd_mfma_both<…, WR = 1>, 4 waves, 16 k per
// chunk, three workgroups per CU): fields of 33–192 heliostats — one to three tiles that are 52–100 % full where a 256-ray
// tile would be 13–75 % — on images and batches large enough to give it a few hundred workgroups.  tools/check_v12.py
// (profiles/r04_k_bwd_tile64.txt), helio_render_bwd per call: B = 256, N = 50, R = 512: 279 → 165 µs; B = 500, N = 128,
// R = 512: 1058 → 631; B = 256, N = 128, R = 256: 150 → 92; B = 500, N = 50, R = 128: 52 → 38.  It loses with four tiles
// (N = 200: the 256-ray tile is 78 % full and stages its slab a quarter as often) and with few workgroups (B = 60,
// N = 50, R = 256: 32 against 22 µs for the small-tile kernel), more of them being needed where the contracted axis is
// short (R <= 128: a workgroup is eight 16-deep chunks).  Same bits as variant 2: a ray's chain over the contracted axis
// does not know how many rays share its tile.
// Larger fields (tools/check_v12.py wide → profiles/r04_m_bwd_tile64_wide.txt, r04_o_bwd_tile64_wide_err90.txt / _err40.txt, and
// the rule table with variant 12 in it): the 64-ray tiles also win wherever the 256-ray tiles (a) pad the field — N = 260
// is two of them, 49 % empty, or one and a 128-ray last tile where the lists split the tails — or (b) leave the chip
// partly idle: B = 4, N = 5000, R = 512 is 320 workgroups for 256 CUs, a second round a quarter full (260 → 202 µs);
// B = 500, N = 260…640, R = 128: 212…316 → 125…246 µs; B = 128, N = 300, R = 256: 162 → 111.  From N = 257 they walk the
// same lists as the 256-ray tiles (the work map in tiles of 64): at err 90 mrad / σs 0.01 B = 128, N = 260, R = 512:
// 425 → 252 µs; B = 500, N = 520, R = 256: 509 → 468.  The rule is a model of both kernels' time in ray SLOTS per image —
// the 256-ray tiles' with their tail form against the 64-ray tiles' — each divided by how full its last round of
// workgroups leaves the chip (256 resp. 768 at a time; a partly filled round counted at 30 % of what it leaves idle).
// Without lists the slots are the field's, padded to whole tiles (a slot of the 64-ray tiles 5 % dearer: their slab is
// staged four times as often).  With lists — which both tilings walk where they pay — the length of a list is not known
// when the kernel is chosen: the EXPECTED padding is what counts, half a last tile (32 rays against 96 with the
// 128-ray tail form, 128 without), the 64-ray slot 8 % dearer: the fine tiles up to N ≈ 770 resp. 1170, the
// 256-ray tiles beyond (configs 4 and 5 stay with them: 2–4 % ahead there).  Against three measured tables (every ray
// live, rules of that moment; every ray live / err 90 mrad, σs 0.01 with lists on both sides: 242 + 62 + 62 sizes) the worst
// misses are 11 %, 8 % and 19 % — the last where the LIVE rays of a 260-heliostat field fit one 256-ray tile, which no
// rule made before the lists exist can know.  More workgroups are needed where the contracted axis is short
// (R <= 128: 1280; else 384).
// (the 64-ray tiles take lists where they are more than one round of the chip — three workgroups per CU — and the
// footprint work carries the three launches in front: ≈20 µs, 5–17 % of a call with every ray live)
static bool tile64_lists_pay(int B, int N, int R) {
    const int ct = (R + (R <= 128 ? 127 : 255)) / (R <= 128 ? 128 : 256);
    return N > 256 && 2l * ct * ((N + 63) / 64) * B > 768 && (long)B * N * R * R >= (1l << 31);
}
// … and for fields of 65–256 heliostats (two to four 64-ray tiles; the 256-ray tiles have nothing to skip there): thousands
// of suns over a small field at the reference's default error scale, where half the rays are dead per image
static bool tile64_lists_small(int B, int N, int R) {
    const int ct = (R + (R <= 128 ? 127 : 255)) / (R <= 128 ? 128 : 256);
    return N > 64 && N <= 256 && 2l * ct * ((N + 63) / 64) * B > 768 && (long)B * N * R * R >= (1l << 31);
}
static bool bwd_tile64(int B, int N, int R) {
    if (N <= 32 || R <= 64) return false;
    const int ct = (R + (R <= 128 ? 127 : 255)) / (R <= 128 ? 128 : 256);
    const long wg64 = 2l * ct * ((N + 63) / 64) * B, wg256 = 2l * ct * ((N + 255) / 256) * B;
    if (N <= 192) return wg64 >= (R <= 128 ? 800 : 240);
    if (wg64 < (R <= 128 ? 1280 : 384)) return false;
    const bool tails = R > 128 && 2l * B * ct >= 512;            // bwd_split_tails: a last tile of at most 128 rays in the 128-ray form
    const int rem = N % 256;
    const bool lists = tile64_lists_pay(B, N, R);
    const double s256 = lists ? N + (tails ? 96.0 : 128.0)
                              : tails ? 256.0 * (N / 256) + (rem == 0 ? 0.0 : rem <= 128 ? 134.0 : 256.0) : 256.0 * ((N + 255) / 256);
    const double s64 = lists ? (N + 32.0) * 1.08 : 64.0 * ((N + 63) / 64) * 1.05;
    auto fill = [](long wgs, long at_a_time) {
        const double rounds = (double)wgs / at_a_time, whole = (double)((wgs + at_a_time - 1) / at_a_time);
        return 1.0 - 0.3 * (1.0 - rounds / whole);
    };
    return s64 / fill(wg64, 768) < s256 / fill(wg256, 256);
}

template <int CT>
__global__ void __launch_bounds__(1024)
render_bwd_fused_small(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                       const float* __restrict__ ys, const float* __restrict__ gimg,
                       const float* __restrict__ helios, const float* __restrict__ sun,
                       const float* __restrict__ action, const float* __restrict__ trig, long trig_b_stride,
                       PlaneK P, const float* __restrict__ g_actual, const float* __restrict__ g_refl,
                       float* __restrict__ g_action, RayLossBwdArgs RL) {
    constexpr int KS = 8 / CT;
    __shared__ float sCc[16 * 64];                   // c coordinates, one private copy per wave
    __shared__ float sRed[16 * 32 * 3];              // [pass][c block][k part][ray][3]
    const int b = blockIdx.y, n0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int pass = wave >> 3, cblk = (wave & 7) / KS, kpart = (wave & 7) % KS;
    const float* __restrict__ G = gimg + (long)b * R * R;

    // the ray adjoint's own inputs (wave 0, one ray per lane of its lower half): in flight during the contraction
    const int n = n0 + lr;
    const long m = (long)b * N + min(n, N - 1);
    RayBwdIn in;
    if (wave == 0) in.load(m, b, min(n, N - 1), helios, sun, action, trig, trig_b_stride, g_actual, g_refl);

    const int pairs = (R + 1) >> 1;
    const int per = (((pairs + KS - 1) / KS) + 1) & ~1;
    const int k_begin = min(R, 2 * per * kpart), k_end = min(R, 2 * per * (kpart + 1));
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    if (64 * cblk < R) {                             // (wave-uniform) c blocks past the image contribute nothing
        if (pass == 0) small_wave_partial<0>(N, R, rays + 4l * b * N, xs, ys, G, 64 * cblk, n0, k_begin, k_end, sCc + wave * 64, m0, m1, m2);
        else small_wave_partial<1>(N, R, rays + 4l * b * N, xs, ys, G, 64 * cblk, n0, k_begin, k_end, sCc + wave * 64, m0, m1, m2);
    }
    if (lh == 0) {
        float* r = sRed + (wave * 32 + lr) * 3;
        r[0] = m0; r[1] = m1; r[2] = m2;
    }
    __syncthreads();
    if (tid >= 32 || n >= N) return;
    // pass 0 → (M0, Ms = My, Mss = Myy), pass 1 → (·, Mt = Mx, Mtt = Mxx); k parts first, then c blocks
    float M0 = 0.f, Mx = 0.f, My = 0.f, Mxx = 0.f, Myy = 0.f;
#pragma unroll
    for (int cb = 0; cb < CT; ++cb) {
        float t[2][3];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int w = 0; w < KS; ++w) {
                const float* r = sRed + (((ps * 8 + cb * KS + w) * 32) + lr) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) t[ps][k] = w == 0 ? r[k] : t[ps][k] + r[k];
            }
        M0 += t[0][0]; Mx += t[1][1]; My += t[0][1]; Mxx += t[1][2]; Myy += t[0][2];
    }
    st3(g_action + 3 * m, geometry_bwd_ray(in, m, n, B, N, P, true, M0, Mx, My, Mxx, Myy, g_actual != nullptr, helios, action, RL));
}

// The sizes the single-launch backward serves (variant 8 forces it; HELIO_BWD_FUSED=0 switches the choice off).
bool render_bwd_is_fused(int B, int N, int R) {
    static const bool off = [] { const char* e = getenv("HELIO_BWD_FUSED"); return e && e[0] == '0'; }();
    if (off || splat_bwd_is_few(B, N, R) || R > 256 || bwd_tile64(B, N, R)) return false;
    // Round 4, from tools/rule_regret.py (device times as HIP-graph replays; profiles/r04_c_rule_regret.txt).  wg = the
    // launch's workgroups, one per 32 rays of a sun, 16 waves each.
    const long wg = (long)B * ((N + 31) / 32);
    // (a) at most 64 of them, R <= 128 — the latency-bound corner, config 3 in it: ON THE DEVICE the two launches are
    //     the shorter (≈9 against ≈12 µs: the 16 waves of a workgroup share one CU's matrix pipe), but a forward +
    //     backward through the Python surface is issue-bound there and one launch less is 17–19.5 against 22–24 µs per
    //     call (config 3, same table): the rule follows the wall clock.  HELIO_BWD_FUSED=0 for graph replays.
    if (R <= 128 && wg <= 64) return true;
    // (b) enough workgroups to fill the chip and few ray blocks per sun: no round trip of the moments, one launch —
    //     B = 256, N = 50, R = 128: 25.0 against 29.5 µs; B = 32, N = 200, R = 128: 13.2 / 16.1; B = 500, N = 32, R = 256:
    //     71.4 / 79.0.  Between (a) and (b) — 64 < wg < 160 — the two launches win (B = 60, N = 50, R = 128: 11.4 / 13.1),
    //     and so they do with 64-pixel images (B = 256, N = 50, R = 64: 12.3 / 15.9) and with many ray blocks.
    if (R > 64 && R <= 128) return wg >= 160 && (N <= 64 || wg <= 512);
    if (R > 128) return N <= 64 && wg >= 256;
    return false;
}

// ideal == nullptr: the render's backward alone; else with the adjoint of HelioEnv.step's two ray losses
bool launch_render_bwd_fused(int B, int N, int R, const float* rays, const float* xs, const float* ys, const float* gimg,
                             const float* helios, const float* sun, const float* action, const float* trig,
                             long trig_b_stride, const helio_plane* plane, const float* g_actual, const float* g_refl,
                             float* g_action, const float* ideal, const float* g_align, const float* g_bound,
                             const float* tp, const float* tn, float W, float H, int exponential_risk, hipStream_t st) {
    if (R > 256 || B > 65535) return false;
    RayLossBwdArgs RL{};
    if (ideal) {
        RL.ideal = ideal; RL.g_align = g_align; RL.g_bound = g_bound;
        RL.g = make_geom(tp, tn, W, H, exponential_risk);
    }
    const dim3 grid((N + 31) / 32, B), block(1024);
    const PlaneK P = to_k(plane);
    if (R <= 64) hipLaunchKernelGGL(render_bwd_fused_small<1>, grid, block, 0, st, B, N, R, rays, xs, ys, gimg, helios, sun, action, trig, trig_b_stride, P, g_actual, g_refl, g_action, RL);
    else if (R <= 128) hipLaunchKernelGGL(render_bwd_fused_small<2>, grid, block, 0, st, B, N, R, rays, xs, ys, gimg, helios, sun, action, trig, trig_b_stride, P, g_actual, g_refl, g_action, RL);
    else hipLaunchKernelGGL(render_bwd_fused_small<4>, grid, block, 0, st, B, N, R, rays, xs, ys, gimg, helios, sun, action, trig, trig_b_stride, P, g_actual, g_refl, g_action, RL);
    return true;
}

// waves per workgroup of the small backward kernel (they split the contracted axis): 8 while that keeps
// the chip at about one wave per SIMD, else 4; HELIO_BWD_KS (4 or 8) forces one — tuning runs only
static int bwd_small_ks(int B, int N, int R) {
    static const int forced = [] { const char* e = getenv("HELIO_BWD_KS"); return e ? atoi(e) : 0; }();
    if (forced == 4 || forced == 8) return forced;
    const long wgs = 2l * B * ((R + 63) / 64) * ((N + 31) / 32);
    return (wgs * 8 <= 1536 && R >= 64) ? 8 : 4;
}

// ray blocks per wave of the small backward kernel (1, 2 or 4); HELIO_BWD_NRB forces one — tuning runs only
static int bwd_small_nrb(int B, int N, int R, int ks) {
    static const int forced = [] { const char* e = getenv("HELIO_BWD_NRB"); return e ? atoi(e) : 0; }();
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    // tools/sweep_bwd_nrb.py: two ray blocks per wave are 3–10 % ahead from N = 300 (B = 25: 36.5 → 35.3 µs at
    // N = 1000, R = 128; 110.9 → 101.8 at R = 256), behind below (N = 100: 7.2 → 8.3 µs); four never pay
    // (344 registers: one wave per SIMD).  Same sums per ray: the bits do not change.
    // … and only while halving the workgroups leaves the chip full (round 4: B = 4, N = 1000, R = 64 — 256 workgroups
    // with one ray block each — 8.5 against 9.8 µs; B = 4, N = 5000, R = 64: 13.7 / 15.1)
    return (N >= 300 && ks == 4 && 2l * B * ((R + 63) / 64) * ((N + 31) / 32) >= 2048) ? 2 : 1;
}

template <int PASS, bool VEC, int WC>
static void launch_bwd_mfma_v(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                              const float* gimg, float* moments, hipStream_t st, CullBwd cull) {
    constexpr int TC = 64 * WC;
    const size_t lds = (64 * ((PASS == 0 ? TC + 4 : TC + 1) + (PASS == 0 ? 260 : 257)) + TC + 64) * sizeof(float);
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(splat_bwd_mfma<PASS, VEC, WC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    const int ct = (R + TC - 1) / TC, nt = (N + 255) / 256;
    const CullBwd c = cull.for_pass(PASS);
    hipLaunchKernelGGL((splat_bwd_mfma<PASS, VEC, WC>), dim3(ct * nt, B), dim3(256 * WC), lds, st, B, N, R, rays, xs, ys, gimg, moments,
                       c.counts, c.idx, c.total, c.map, c.ct);
}

template <bool VEC, int WC, int WR, bool DB = false>
static void launch_bwd_mfma_both_v(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                                   const float* gimg, float* moments, hipStream_t st, CullBwd c) {
    constexpr int TC = 64 * WC, T = 64 * WR, KC = WR == 1 ? 16 : ((WR == 2 || DB) ? 32 : 64);
    const size_t lds = ((DB ? 2 : 1) * KC * ((TC + 4) + (T + 4)) + TC + 64) * sizeof(float);      // pass 0's pitches: the larger of the two
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(splat_bwd_mfma_both<VEC, WC, WR, DB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    const int ct = (R + TC - 1) / TC, nt = (N + 255) / 256;
    if (WR == 4) {
        hipLaunchKernelGGL((splat_bwd_mfma_both<VEC, WC, WR, DB>), dim3(ct * nt, B, 2), dim3(64 * WC * WR), lds, st, B, N, R, rays, xs, ys, gimg,
                           moments, c.counts, c.idx, c.total, c.map, c.ct, c.set_lists, c.set_items, CULL_BWD_TILE);
    } else if (WR == 1 && c.counts) {
        // the 64-ray tiles over lists: the map (cull.hip) counts tiles of 64 rays, no tail form
        hipLaunchKernelGGL((splat_bwd_mfma_both<VEC, WC, WR, DB>), dim3(ct * ((N + T - 1) / T), B, 2), dim3(64 * WC * WR), lds, st, B, N, R, rays, xs, ys, gimg,
                           moments, c.counts, c.idx, c.total, c.map, c.ct, c.set_lists, c.set_items, T);
    } else if (!c.counts) {
        // dense launches of the narrow forms, tiles numbered in T rays: WR = 1 (variant 12) — and the HELIO_BWD_WR2
        // experiment, the 128-ray form over EVERY tile
        hipLaunchKernelGGL((splat_bwd_mfma_both<VEC, WC, WR, DB>), dim3(ct * ((N + T - 1) / T), B, 2), dim3(64 * WC * WR), lds, st, B, N, R, rays, xs, ys, gimg,
                           moments, nullptr, nullptr, nullptr, nullptr, 1, 0l, 0l, T);
    } else {
        // the short last tiles: at most one per list (and c tile, where a list serves all of an image's c tiles)
        const long items = c.set_lists * (c.ct > 1 ? 1 : ct);
        hipLaunchKernelGGL((splat_bwd_mfma_both<VEC, WC, WR, DB>), dim3((unsigned)items, 1, 2), dim3(64 * WC * WR), lds, st, B, N, R, rays, xs, ys,
                           gimg, moments, c.counts, c.idx, c.tail_total, c.tail_map, c.ct, c.set_lists, c.set_lists, CULL_BWD_TILE);
    }
}

template <int PASS>
static void launch_bwd_mfma(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                            const float* gimg, float* moments, hipStream_t st, CullBwd cull) {
    const bool narrow = R <= 128;          // 128-wide c tiles: a 256-wide one would be at least half padding
    if constexpr (PASS == 0) {
        // 16-byte staging of the grad-image slab needs whole 4-pixel pieces per row (the image base is
        // 16-byte aligned by the ABI's contract)
        if ((R & 3) == 0) return narrow ? launch_bwd_mfma_v<0, true, 2>(B, N, R, rays, xs, ys, gimg, moments, st, cull)
                                        : launch_bwd_mfma_v<0, true, 4>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        return narrow ? launch_bwd_mfma_v<0, false, 2>(B, N, R, rays, xs, ys, gimg, moments, st, cull)
                      : launch_bwd_mfma_v<0, false, 4>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
    } else {
        return narrow ? launch_bwd_mfma_v<1, false, 2>(B, N, R, rays, xs, ys, gimg, moments, st, cull)
                      : launch_bwd_mfma_v<1, false, 4>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
    }
}

// both passes in one launch (HELIO_BWD_PASSES=2: two, for A/B runs)
static void launch_bwd_mfma_both(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                                 const float* gimg, float* moments, hipStream_t st, CullBwd cull) {
    static const bool two_launches = [] { const char* e = getenv("HELIO_BWD_PASSES"); return e && e[0] == '2'; }();   // A/B runs
    const bool vec = (R & 3) == 0;
    if (two_launches) {
        launch_bwd_mfma<0>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        launch_bwd_mfma<1>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
    } else if (R <= 128 && !cull.counts && [] { const char* e = getenv("HELIO_BWD_WR2"); return e && e[0] == '1'; }()) {
        // experiment (round 4): 128 c × 128 rays, 4 waves, 34 KB of LDS — several workgroups per CU, each in another's
        // producer phase — over EVERY tile of a dense launch
        if (vec) launch_bwd_mfma_both_v<true, 2, 2>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        else launch_bwd_mfma_both_v<false, 2, 2>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
    } else if (R <= 128) {
        // 128-wide c tiles: one 8-wave workgroup per CU either way (101 KB of LDS), so pass 0 loses nothing by running
        // at pass 1's register count
        if (vec) launch_bwd_mfma_both_v<true, 2, 4>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        else launch_bwd_mfma_both_v<false, 2, 4>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
    } else if (vec && !cull.counts && [] { const char* e = getenv("HELIO_BWD_WR2"); return e && e[0] == '1'; }()) {
        launch_bwd_mfma_both_v<true, 4, 2>(B, N, R, rays, xs, ys, gimg, moments, st, cull);       // experiment: 128-ray tiles, two workgroups per CU
    } else if (vec) {
        // HELIO_BWD_DB=1: the double-buffered body — the same bits, measured SLOWER at config 4 (dense 8.71 against 8.29 ms,
        // with the lists 5.28 / 5.08; profiles/r04_e_bwd_db.txt) and therefore not the default: kept as the record of the
        // experiment and for the test that holds the two bodies to the same bits
        const char* e = getenv("HELIO_BWD_DB");        // (read per call — these launches are milliseconds — so that a test can compare the two bodies in one process)
        const bool db = e && e[0] == '1';
        if (db) launch_bwd_mfma_both_v<true, 4, 4, true>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        else launch_bwd_mfma_both_v<true, 4, 4>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        if (cull.tail_map) launch_bwd_mfma_both_v<true, 4, 2>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
    } else {
        launch_bwd_mfma_both_v<false, 4, 4>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        if (cull.tail_map) launch_bwd_mfma_both_v<false, 4, 2>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
    }
}

// the last tile of a list in the 128-ray form when it holds at most 128 rays (cull.h): only with the one-launch form of
// the 256-wide tiles, and only where the short tiles — about every other list's — can fill the chip: their launch
// runs AFTER the 256-ray launch, so a handful of them is a workgroup's duration of pure latency instead of a place in
// the last round of that launch (B = 32, N = 5000, R = 256 at err 90: 64 lists, 312 → 368 µs; B = 256, N = 1000,
// R = 256: 512 lists, 587 → 460 µs; config 4: 4096 lists).  HELIO_BWD_TAIL=0 switches it off (A/B runs).
static bool bwd_split_tails(int R, long lists_both_passes) {
    static const bool off = [] { const char* e = getenv("HELIO_BWD_TAIL"); return e && e[0] == '0'; }();
    static const bool two_launches = [] { const char* e = getenv("HELIO_BWD_PASSES"); return e && e[0] == '2'; }();
    return R > 128 && !off && !two_launches && lists_both_passes >= 512;
}

// ----------------------------------------------------------------------------------------------
// Split-bf16 backward (opt-in, variant 5): the two contractions above on the bf16 matrix pipe, with
// the scheme of splat_fwd_mfma_bf16x3 (splat_fwd.hip): both operands — the grad-image values and
// the factors — are split exactly into three bf16 pieces and a product is issued as six of its
// nine partial products on v_mfma_f32_32x32x16_bf16, f32 accumulation; what is dropped is below
// 2^-23 of each product.  Workgroup = 8 waves = 256 c × 256 rays (2×4 waves of 128×64); trip = 16
// of the contracted axis; waves 0-3 stage the grad-image slab (thread ↔ c, 16 k of the trip,
// prefetched a trip ahead), waves 4-7 the factor table (thread ↔ ray); the two waves of a SIMD
// (w, w+4) run "MFMAs of this trip" and "operands of the next" in opposite order.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack_hi16(float lo_elem, float hi_elem) {
    return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}

// v[0..7] (f32) → three uint4 of bf16 pieces (hi, mid, lo), exact
__device__ __forceinline__ void split3(const float* v, uint4& h, uint4& m, uint4& l) {
    float hi[8], mid[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = __uint_as_float(__float_as_uint(v[j]) & 0xFFFF0000u);
        const float r1 = v[j] - hi[j];
        mid[j] = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
        lo[j] = r1 - mid[j];
    }
    h = make_uint4(pack_hi16(hi[0], hi[1]), pack_hi16(hi[2], hi[3]), pack_hi16(hi[4], hi[5]), pack_hi16(hi[6], hi[7]));
    m = make_uint4(pack_hi16(mid[0], mid[1]), pack_hi16(mid[2], mid[3]), pack_hi16(mid[4], mid[5]), pack_hi16(mid[6], mid[7]));
    l = make_uint4(pack_hi16(lo[0], lo[1]), pack_hi16(lo[2], lo[3]), pack_hi16(lo[4], lo[5]), pack_hi16(lo[6], lo[7]));
}

template <int PASS>
__global__ void __launch_bounds__(512)
splat_bwd_mfma_bf16x3(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                      const float* __restrict__ ys, const float* __restrict__ gimg, float* __restrict__ moments) {
    constexpr int KC = 16, T = 256;
    constexpr int PIECE = 2 * T * 16;                 // bytes: [k-half][c or ray][8 × bf16]
    constexpr int TABLE = 3 * PIECE, BUF = 2 * TABLE; // grad-image table, then factor table
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    float* sCc = reinterpret_cast<float*>(ldsb + 2 * BUF);

    const int c_tiles = (R + T - 1) / T;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int c0 = (blockIdx.x % c_tiles) * T, n0 = (blockIdx.x / c_tiles) * T;
    const int wi = (wave >> 2) * 128, wj = (wave & 3) * 64;
    const float* __restrict__ ccoord = PASS == 0 ? ys : xs;   // coordinates along c
    const float* __restrict__ kcoord = PASS == 0 ? xs : ys;   // coordinates along k
    const float* __restrict__ G = gimg + (long)b * R * R;

    if (tid < T) sCc[tid] = ccoord[min(c0 + tid, R - 1)];

    const bool grole = tid < T;                      // waves 0-3: grad-image slab; waves 4-7: factor table
    const int pp = grole ? tid : tid - T;            // c of the tile, or ray of the tile
    float4 q = make_float4(0.f, 0.f, 1.f, 1e30f);
    if (!grole && n0 + pp < N) q = reinterpret_cast<const float4*>(rays)[(long)b * N + n0 + pp];
    const float sk = __builtin_sqrtf(q.z);
    const float fshift = (PASS == 0 ? q.x : q.y) * sk;
    const float fcc = PASS == 0 ? q.w * q.z : ((!grole && n0 + pp < N) ? 0.0f : 1e30f);
    const int ptab = (grole ? 0 : TABLE) + pp * 16;
    const bool vec_ok = (R & 3) == 0;

    float gv[16];
    auto load_slab = [&](int k0) {                   // Gm[k][c], 16 k of the trip, for this thread's c
        if (PASS == 0) {                             // Gm[k][c] = G[k0+k][c0+c]: lanes ↔ c, coalesced rows
            const int col = c0 + pp;
#pragma unroll
            for (int j = 0; j < 16; ++j) gv[j] = (k0 + j < R && col < R) ? G[(long)(k0 + j) * R + col] : 0.0f;
        } else {                                     // Gm[k][c] = G[c0+c][k0+k]: 64 contiguous bytes per lane
            const int row = c0 + pp;
            if (vec_ok && row < R && k0 + 15 < R) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float4 t = *reinterpret_cast<const float4*>(G + (long)row * R + k0 + 4 * v);
                    gv[4 * v] = t.x; gv[4 * v + 1] = t.y; gv[4 * v + 2] = t.z; gv[4 * v + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) gv[j] = (row < R && k0 + j < R) ? G[(long)row * R + k0 + j] : 0.0f;
            }
        }
    };
    auto produce = [&](int k0, int buf) {            // this thread's 16 operand values of the trip at k0
        unsigned char* base = ldsb + buf * BUF + ptab;
        float v[16];
        if (grole) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = gv[j];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float t = __builtin_fmaf(kcoord[min(k0 + j, R - 1)], sk, fshift);      // wave-uniform coordinate
                v[j] = (k0 + j < R) ? __builtin_amdgcn_exp2f(-__builtin_fmaf(t, t, fcc)) : 0.0f;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 ph, pm, pl;
            split3(v + 8 * h, ph, pm, pl);
            *reinterpret_cast<uint4*>(base + 0 * PIECE + h * T * 16) = ph;
            *reinterpret_cast<uint4*>(base + 1 * PIECE + h * T * 16) = pm;
            *reinterpret_cast<uint4*>(base + 2 * PIECE + h * T * 16) = pl;
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][nb][e] = 0.0f;

    const int trips = (R + KC - 1) / KC;
    if (grole) load_slab(0);
    produce(0, 0);
    if (grole) load_slab(KC);                        // (past the image: zeros)
    const int offG = (lh * T + wi + lr) * 16, offF = TABLE + (lh * T + wj + lr) * 16;
    for (int t = 0; t < trips; ++t) {
        __syncthreads();                             // tables[t&1] complete
        const int buf = t & 1;
        auto consume = [&]() {
            const unsigned char* tb = ldsb + buf * BUF;
            bf16x8 fp[3][2];
#pragma unroll
            for (int P = 0; P < 3; ++P)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) fp[P][nb] = *reinterpret_cast<const bf16x8*>(tb + offF + P * PIECE + nb * 512);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                bf16x8 gp[3];
#pragma unroll
                for (int P = 0; P < 3; ++P) gp[P] = *reinterpret_cast<const bf16x8*>(tb + offG + P * PIECE + cb * 512);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    f32x16 v = acc[cb][nb];
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gp[1], fp[1][nb], v, 0, 0, 0);      // small terms first
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gp[2], fp[0][nb], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gp[0], fp[2][nb], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gp[1], fp[0][nb], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gp[0], fp[1][nb], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gp[0], fp[0][nb], v, 0, 0, 0);
                    acc[cb][nb] = v;
                }
            }
        };
        auto next = [&]() {                          // operands of trip t+1 (the last trip stages zeros)
            produce((t + 1) * KC, buf ^ 1);
            if (grole) load_slab((t + 2) * KC);
        };
        if (wave & 4) {
            next();
            __builtin_amdgcn_sched_barrier(0);
            consume();
        } else {
            consume();
            __builtin_amdgcn_sched_barrier(0);
            next();
        }
    }

    // epilogue: this lane's ray is column lr of ray block nb; the 16 registers of an accumulator
    // block are c = (e&3) + 8(e>>2) + 4·lh of c block cb; a wave covers two 64-wide partial blocks
    const int JB = (R + 63) / 64;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = n0 + wj + 32 * nb + lr;
        float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) h = reinterpret_cast<const float4*>(rays)[(long)b * N + n];
        const float hshift = PASS == 0 ? h.y : h.x;
        const float hcc = PASS == 0 ? 0.0f : h.w;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int cb = 2 * half + cc;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int cl = wi + 32 * cb + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float s = sCc[cl] + hshift;
                    const float w = __builtin_amdgcn_exp2f(-(__builtin_fmaf(s, s, hcc) * h.z)) * acc[cb][nb][e];
                    m0 += w;
                    m1 = __builtin_fmaf(s, w, m1);
                    m2 = __builtin_fmaf(s * s, w, m2);
                }
            }
            m0 += __shfl_xor(m0, 32); m1 += __shfl_xor(m1, 32); m2 += __shfl_xor(m2, 32);
            const int cstart = c0 + wi + 64 * half;
            if (lh == 0 && n < N && cstart < R) {
                float* o = moments + (((long)b * JB + cstart / 64) * N + n) * HELIO_MOMENT_STRIDE;
                if (PASS == 0) { o[0] = m0; o[2] = m1; o[4] = m2; }
                else { o[1] = m1; o[3] = m2; }
            }
        }
    }
}

template <int PASS>
static void launch_bwd_bf16x3(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                              const float* gimg, float* moments, hipStream_t st) {
    const size_t lds = 2 * (2 * 3 * 2 * 256 * 16) + 256 * sizeof(float);
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(splat_bwd_mfma_bf16x3<PASS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    const int ct = (R + 255) / 256, nt = (N + 255) / 256;
    hipLaunchKernelGGL(splat_bwd_mfma_bf16x3<PASS>, dim3(ct * nt, B), dim3(512), lds, st, B, N, R, rays, xs, ys, gimg, moments);
}

// Few rays per image (the reference's test-time-compute sweeps run ONE heliostat and 500 suns,
// run_experiments.py:31-56): the 64-ray tiles of the kernels above would be ≥ 90 % padding and the
// pass is bound by streaming grad_image once.  A workgroup owns 64 image columns of one sun
// (lanes ↔ columns, waves ↔ rows i ≡ wave mod 4) and NR rays in registers; per (pixel, ray) it
// evaluates only the row factor A — the column factor E is constant per lane and multiplies
// the finished sums.  Partials have the same [b, column block, ray, 5] layout.
// FUSED: grad_image is not read but formed on the fly from HelioEnv.step's loss block
// (loss_grad_pixel on img / target / distance map): backward through mse and dist without ever
// materialising the [B,R,R] cotangent.
template <int NR, bool FUSED>
__global__ void __launch_bounds__(256)
splat_bwd_few(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
              const float* __restrict__ ys, const float* __restrict__ gimg, LossGradArgs L,
              float* __restrict__ moments) {
    __shared__ float sRed[4][NR][5];
    const int jb = blockIdx.x, b = blockIdx.y, n0 = blockIdx.z * NR;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = jb * 64 + lane;
    const bool col_ok = j < R;
    const long base = (long)b * R * R + (col_ok ? j : 0);

    float qa[NR], qk[NR], qc[NR], sj[NR], ej[NR];
    const float yj = ys[min(j, R - 1)];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + r < N) q = reinterpret_cast<const float4*>(rays)[(long)b * N + n0 + r];      // wave-uniform
        qa[r] = q.x; qk[r] = q.z; qc[r] = q.w;
        sj[r] = yj + q.y;
        ej[r] = col_ok ? __builtin_amdgcn_exp2f(-((sj[r] * sj[r]) * q.z)) : 0.0f;
    }
    float ls = 1.0f, km = 0.0f, kd = 0.0f;
    if constexpr (FUSED) L.constants(b, B, (long)R * R, ls, km, kd);

    float m0[NR], mt[NR], mtt[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { m0[r] = 0.f; mt[r] = 0.f; mtt[r] = 0.f; }

    constexpr int U = 8;                      // rows in flight per lane
    for (int i0 = wave; i0 < R; i0 += 4 * U) {
        float g[U], xi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 4 * u;
            const bool ok = col_ok && i < R;
            const long p = base + (long)min(i, R - 1) * R;
            xi[u] = xs[min(i, R - 1)];
            if constexpr (FUSED) {
                const float v = loss_grad_pixel(L.img[p], L.target[p], L.dmaps[p], ls, km, kd);
                g[u] = ok ? v : 0.0f;
            } else {
                g[u] = ok ? gimg[p] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float t = xi[u] + qa[r];
                const float w = __builtin_amdgcn_exp2f(-(__builtin_fmaf(t, t, qc[r]) * qk[r])) * g[u];
                m0[r] += w;
                mt[r] = __builtin_fmaf(t, w, mt[r]);
                mtt[r] = __builtin_fmaf(t * t, w, mtt[r]);
            }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float v[5] = {ej[r] * m0[r], ej[r] * mt[r], sj[r] * (ej[r] * m0[r]), ej[r] * mtt[r],
                      (sj[r] * sj[r]) * (ej[r] * m0[r])};            // M0, Mt, Ms, Mtt, Mss
#pragma unroll
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d);
            if (lane == 0) sRed[wave][r][k] = v[k];
        }
    }
    __syncthreads();
    if (tid < NR * 5) {
        const int r = tid / 5, k = tid % 5;
        if (n0 + r < N) {
            const int JB = (R + 63) / 64;
            moments[(((long)b * JB + jb) * N + n0 + r) * HELIO_MOMENT_STRIDE + k] =
                (sRed[0][r][k] + sRed[1][r][k]) + (sRed[2][r][k] + sRed[3][r][k]);
        }
    }
}

// The same with 16-byte lanes for R % 4 == 0: a wave reads 4 rows × 64 columns per load
// instruction (lane = row-in-group × 16 + column quad), so the stream runs at the rate of the
// float4 loss kernels instead of 256-byte row segments.
template <int NR, bool FUSED>
__global__ void __launch_bounds__(256)
splat_bwd_few_vec(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                  const float* __restrict__ ys, const float* __restrict__ gimg, LossGradArgs L,
                  float* __restrict__ moments) {
    __shared__ float sRed[4][NR][5];
    const int jb = blockIdx.x, b = blockIdx.y, n0 = blockIdx.z * NR;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = lane >> 4, j = jb * 64 + 4 * (lane & 15);           // R % 4 == 0: a quad is in or out as a whole
    const bool col_ok = j < R;
    const long base = (long)b * R * R + (col_ok ? j : 0);

    float qa[NR], qk[NR], qc[NR], sj[NR][4], ej[NR][4];
    float yj[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) yj[c] = ys[min(j + c, R - 1)];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + r < N) q = reinterpret_cast<const float4*>(rays)[(long)b * N + n0 + r];
        qa[r] = q.x; qk[r] = q.z; qc[r] = q.w;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sj[r][c] = yj[c] + q.y;
            ej[r][c] = col_ok ? __builtin_amdgcn_exp2f(-((sj[r][c] * sj[r][c]) * q.z)) : 0.0f;
        }
    }
    float ls = 1.0f, km = 0.0f, kd = 0.0f;
    if constexpr (FUSED) L.constants(b, B, (long)R * R, ls, km, kd);

    float m0[NR][4], mt[NR][4], mtt[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { m0[r][c] = 0.f; mt[r][c] = 0.f; mtt[r][c] = 0.f; }

    constexpr int U = NR == 4 ? 2 : 4;          // 16-row groups in flight per lane
    for (int i0 = 4 * wave + sub; i0 < R; i0 += 16 * U) {
        float4 g[U];
        float xi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 16 * u;
            const bool ok = col_ok && i < R;
            const long p = base + (long)min(i, R - 1) * R;
            xi[u] = xs[min(i, R - 1)];
            if constexpr (FUSED) {
                const float4 a = *reinterpret_cast<const float4*>(L.img + p);
                const float4 t = *reinterpret_cast<const float4*>(L.target + p);
                const float4 d = *reinterpret_cast<const float4*>(L.dmaps + p);
                g[u] = make_float4(loss_grad_pixel(a.x, t.x, d.x, ls, km, kd), loss_grad_pixel(a.y, t.y, d.y, ls, km, kd),
                                   loss_grad_pixel(a.z, t.z, d.z, ls, km, kd), loss_grad_pixel(a.w, t.w, d.w, ls, km, kd));
            } else {
                g[u] = *reinterpret_cast<const float4*>(gimg + p);
            }
            if (!ok) g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float gv[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float t = xi[u] + qa[r];
                const float a = __builtin_amdgcn_exp2f(-(__builtin_fmaf(t, t, qc[r]) * qk[r]));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float w = a * gv[c];
                    m0[r][c] += w;
                    mt[r][c] = __builtin_fmaf(t, w, mt[r][c]);
                    mtt[r][c] = __builtin_fmaf(t * t, w, mtt[r][c]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};                      // M0, Mt, Ms, Mtt, Mss
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float e0 = ej[r][c] * m0[r][c];
            v[0] += e0; v[1] += ej[r][c] * mt[r][c]; v[2] += sj[r][c] * e0; v[3] += ej[r][c] * mtt[r][c];
            v[4] += (sj[r][c] * sj[r][c]) * e0;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d);
            if (lane == 0) sRed[wave][r][k] = v[k];
        }
    }
    __syncthreads();
    if (tid < NR * 5) {
        const int r = tid / 5, k = tid % 5;
        if (n0 + r < N) {
            const int JB = (R + 63) / 64;
            moments[(((long)b * JB + jb) * N + n0 + r) * HELIO_MOMENT_STRIDE + k] =
                (sRed[0][r][k] + sRed[1][r][k]) + (sRed[2][r][k] + sRed[3][r][k]);
        }
    }
}

template <bool FUSED>
static void launch_bwd_few(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                           const float* gimg, const LossGradArgs& L, float* moments, hipStream_t st) {
    const int JB = (R + 63) / 64;
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (R & 3) == 0 && (FUSED ? (a16(L.img) && a16(L.target) && a16(L.dmaps)) : a16(gimg));
#define HELIO_FEW(K, NRV, GZ) hipLaunchKernelGGL((K<NRV, FUSED>), dim3(JB, B, GZ), dim3(256), 0, st, B, N, R, rays, xs, ys, gimg, L, moments)
    if (vec) {
        if (N == 1) HELIO_FEW(splat_bwd_few_vec, 1, 1);
        else if (N == 2) HELIO_FEW(splat_bwd_few_vec, 2, 1);
        else HELIO_FEW(splat_bwd_few_vec, 4, (N + 3) / 4);
    } else {
        if (N == 1) HELIO_FEW(splat_bwd_few, 1, 1);
        else if (N == 2) HELIO_FEW(splat_bwd_few, 2, 1);
        else HELIO_FEW(splat_bwd_few, 4, (N + 3) / 4);
    }
#undef HELIO_FEW
}

// where the few-ray kernel wins (tools/sweep_bwd.py, MI355X): its time grows with B·N·R², the
// MFMA kernels' with the number of 64-ray tiles
bool splat_bwd_is_few(int B, int N) { return N <= 8 || (N <= 16 && B <= 64) || (N <= 32 && B <= 8); }
// … and, knowing the image size, the rule the render's backward uses (round 4, tools/rule_regret.py,
// profiles/r04_c_rule_regret.txt): with 3..8 rays per image and few pixels in all the small MFMA kernel's latency is the
// shorter one — N = 8, R = 128, B <= 60: 11.7 against 8.8–9.4 µs; from ≈2 M pixels (B = 60, R = 256) the streaming kernel
// is level or ahead.  One or two rays: always.  (The two-argument form stays the env step's: its few-ray path also
// forms the image cotangent on the fly.)
bool splat_bwd_is_few(int B, int N, int R) {
    return N <= 8 && (N <= 2 || (long)B * R * R >= (1l << 21));       // (9..32 rays: never ahead in the table — B = 60, N = 16, R = 256: 20.7 against 16.7 µs)
}

// backward through the image losses with the cotangent formed on the fly (few rays only)
void launch_splat_bwd_fused_loss(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                                 const LossGradArgs& L, float* moments, hipStream_t st) {
    launch_bwd_few<true>(B, N, R, rays, xs, ys, nullptr, L, moments, st);
}

void launch_splat_bwd_fused_loss_raw(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                                     const float* img, const float* target, const float* tx, const float* dmaps,
                                     const float* keep, const float* g_mse, const float* g_dist, float* moments,
                                     hipStream_t st) {
    launch_splat_bwd_fused_loss(B, N, R, rays, xs, ys, LossGradArgs{img, target, tx, dmaps, keep, g_mse, g_dist}, moments, st);
}

int splat_bwd_blocks(int R) { return (R + BW_JT - 1) / BW_JT; }

// variant: 0 = by problem size, 1 = VALU kernel, 2 = MFMA kernels (256-tiles), 3 = MFMA small tiles,
// 4 = few-ray streaming kernel, 5 = split-bf16 MFMA kernels (opt-in), 6 / 7 = the small kernel with 4 / 8 waves,
// 9 / 10 / 11 = its forms by the cut of the contracted axis, 12 = the LDS-tile kernel in 64-ray tiles
// the kernel family variant 0 stands for at this size
static int splat_bwd_choice(int B, int N, int R) {
    if (splat_bwd_is_few(B, N, R)) return 4;
    if (bwd_tile64(B, N, R)) return 12;
    // tools/sweep_bwd.py, tools/sweep_bwd_mid.py: the LDS-tile kernels (256 rays × 256 c, or × 128 c for images of at
    // most 128 pixels across) against the small-tile kernel; below 65 pixels even the narrow tile is half padding and
    // the small-tile kernel is its equal.  Both passes of the LDS-tile kernels are ONE launch of 2·tiles workgroups, one
    // per CU at a time, so its duration is a step function of its rounds of 256 (23 / 69 / 130 µs a round at R = 128 /
    // 256 / 512) while the small-tile kernel's grows with the work: the LDS tiles win where their rounds are well
    // filled — by workgroups and, inside a tile, by rays (N = 300 is two ray tiles, the second 17 % full).  Measured
    // crossover: a fill of ≈0.6 (tiles = 80 of N ≥ 1000: 23 against 29 µs at R = 128, 69 / 80 at 256, 132 / 145 at 512;
    // tiles = 64: 23 / 20, 69 / 54, 130 / 103; N = 300, tiles = 192: 44 / 38, 136 / 118, 260 / 207).
    if (R <= 64 || N < 96) return 3;
    const long ray_tiles = (N + 255) / 256, tiles = (long)B * (R <= 128 ? 1 : (R + 255) / 256) * ray_tiles;
    if (tiles >= 512) return 2;
    const long rounds = (2 * tiles + 255) / 256;
    const double fill = (double)(2 * tiles) / (256.0 * rounds) * N / (256.0 * ray_tiles);
    // (two rounds of which the second is less than half full — B = 4, N = 5000, R = 512: 320 workgroups — go to the
    // small-tile kernel whatever the fill says: 259 against 314–321 µs, tools/rule_regret.py)
    if (rounds == 2 && 2 * tiles < 384) return 3;
    return fill >= 0.58 ? 2 : 3;
}

// the small-tile kernel's form without a split of the contracted axis (4 independent waves of 64 rays each): from
// N = 600 with workgroups enough (tools/sweep_bwd_nrb.py)
static bool small_whole_k(int B, int N, int R) {
    static const int ks_exp = [] { const char* e = getenv("HELIO_BWD_KSX"); return e ? atoi(e) : 0; }();     // tuning runs
    // round 4 (tools/rule_regret.py): what counts is how full the 256-ray groups are — N = 200 (78 %) gains like N = 1000
    // (B = 256, N = 200, R = 64: 21.7 against 28.2 µs; B = 60, N = 200, R = 256: 59.7 / 65.6), N = 300 (59 %) loses — and
    // that there are waves enough to hide its unstaged grad-image loads: 400 workgroups where the contracted axis is
    // long (B = 4, N = 1000, R = 512, 256 workgroups: 72.5 against 63.7 µs), 256 where it is 64
    const long groups = (N + 255) / 256, wgs = 2l * ((R + 63) / 64) * groups * B;
    const bool full = N >= 600 || 4l * N >= 3l * 256 * groups;
    return ks_exp == 0 && full && wgs >= (R <= 64 ? 256 : 400);
}

// The forms of the small-tile kernel that differ in their BITS are the ways its contracted axis is cut: 9 = not at
// all (the whole-k form), 10 = between four waves, 11 = between eight (ray blocks per wave, lists and workgroup shape
// change which wave holds a ray, never the order of its sums).  Variant 3 stands for the one the size rules pick.
static int small_form(int B, int N, int R) { return small_whole_k(B, N, R) ? 9 : bwd_small_ks(B, N, R) == 8 ? 11 : 10; }

// What helio_render_bwd's variant 0 resolves to at (B, N, R), as a variant that pins every choice the bits of the
// gradient depend on (include/helio.h, helio_render_bwd_choice): a shard of a batch that passes the WHOLE batch's
// choice gets the unsharded gradient's rows bit for bit.
int render_bwd_choice(int B, int N, int R) {
    if (render_bwd_is_fused(B, N, R)) return 8;
    const int v = splat_bwd_choice(B, N, R);
    return v == 3 ? small_form(B, N, R) : v;
}

// variant → the kernel family and form it stands for at this size (0, 3 and 8 are resolved by the size rules)
static int resolve_bwd(int variant, int B, int N, int R) {
    if (variant == 8) variant = 3;       // 8 is helio_render_bwd's single-launch form; its moments alone are the small kernel's
    if (variant == 0) variant = splat_bwd_choice(B, N, R);
    if (variant == 3) variant = small_form(B, N, R);
    return variant;
}

// Skipping rays whose footprint is identically zero on the image (cull.h): the LDS-tile kernels walk 256-ray
// tiles of a per-image list, the small-tile kernel in its whole-k form (few images of many heliostats) 256-ray
// groups of it.  In the backward a list means FEWER workgroups, not shorter ones (a ray's moments are a sum over
// the whole image), so it pays only where the dense grid is more than one round of the chip: measured at B = 4,
// N = 5000, R = 512 — 160 tiles, one round — the two passes take 134 µs each with or without the list and the
// three small launches in front of them cost 21 µs; at B = 16, R = 256 (320 tiles): 291 → 177 µs
// (tools/bench_few_images.py).
// → the kernel this call runs CAN walk a list (what a launch with enough scratch does) …
static bool cull_bwd_possible(int variant, int B, int N, int R) {
    variant = resolve_bwd(variant, B, N, R);
    // (up to 256 rays the 256-ray tiles and the small tiles' 256-ray groups are one per image: nothing to skip; the 64-ray
    // tiles are several from 65 rays)
    if (!cull_enabled() || N <= (variant == 12 ? 64 : 256)) return false;
    return variant == 2 || variant == 9 || variant == 12;
}
// … and it PAYS (what the size query answers: a caller that sizes its scratch by the query hands none otherwise)
static bool cull_bwd_wanted(int variant, int B, int N, int R) {
    variant = resolve_bwd(variant, B, N, R);
    if (!cull_bwd_possible(variant, B, N, R)) return false;
    const long ray_tiles = (N + 255) / 256;
    if (variant == 2) return 2l * B * ((R + (R <= 128 ? 127 : 255)) / (R <= 128 ? 128 : 256)) * ray_tiles > 256;   // (both passes are one launch)
    // the 64-ray tiles: three workgroups per CU, and footprint work enough beside the three launches in front
    if (variant == 12) return N > 256 ? tile64_lists_pay(B, N, R) : tile64_lists_small(B, N, R);
    // (both passes in one launch; two 4-wave workgroups fit a CU; and enough footprint work for the ≈15 µs of the
    // launches in front to be small beside it — grid A/B at err 90 / σs 0.01: B = 32, N = 5000, R = 64, 45 µs dense:
    // 54 µs with lists; B = 4, N = 5000, R = 256, 85 µs: 72 µs; B = 256, N = 5000, R = 64, 325 µs: 271 µs)
    // (tried: the listed rays' parameters compacted beside the indices, so that a workgroup's first load does not wait
    // for another — no difference anywhere on the grid; the cost of a list with every ray live is its three launches)
    return N >= 1024 && 2l * B * ((R + 63) / 64) * ray_tiles > 512 && (long)B * N * R * R >= (1l << 30);
}

// lists per image and pass: the LDS-tile kernels' c tiles where an image is 2..8 of them wide (cull.h), else 1
static int cull_bwd_ct(int variant, int B, int N, int R) {
    variant = resolve_bwd(variant, B, N, R);
    const int c_tiles = (R + 255) / 256;
    return (variant == 2 || variant == 12) && R > 128 && c_tiles > 1 && c_tiles <= CULL_BWD_MAX_CT ? c_tiles : 1;
}

long splat_bwd_scratch_bytes(int B, int N, int R, int variant) {
    return cull_bwd_wanted(variant, B, N, R) ? cull_bwd_bytes(B, N, cull_bwd_ct(variant, B, N, R), resolve_bwd(variant, B, N, R) == 12 ? 64 : CULL_BWD_TILE) : 0;
}

int launch_splat_bwd(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                     const float* gimg, float* moments, int variant, void* scratch, long scratch_bytes, hipStream_t st) {
    const bool by_rule = variant == 0 || variant == 3 || variant == 8;      // (the tuning switches below act on the rules' own choice only)
    variant = resolve_bwd(variant, B, N, R);
    if (variant == 5) {
        launch_bwd_bf16x3<0>(B, N, R, rays, xs, ys, gimg, moments, st);
        launch_bwd_bf16x3<1>(B, N, R, rays, xs, ys, gimg, moments, st);
        return HELIO_OK;
    }
    if (variant == 4) {
        if ((N + 3) / 4 > 65535) return HELIO_E_INVALID;          // grid.z; never chosen for that many rays
        launch_bwd_few<false>(B, N, R, rays, xs, ys, gimg, LossGradArgs{}, moments, st);
        return HELIO_OK;
    }
    if (variant == 6 || variant == 7 || (variant >= 9 && variant <= 11)) {
        // 9 / 10 / 11: the small kernel's forms (small_form above: what 3 resolves to); 6 / 7: four / eight waves and
        // ONE ray block per wave whatever the size (tests, tuning)
        const int ct = (R + 63) / 64, nt = (N + 31) / 32;
        const int ks = (variant == 7 || variant == 11) ? 8 : 4;
        const int nrb = variant >= 9 ? bwd_small_nrb(B, N, R, ks) : 1;
        static const int ks_exp_env = [] { const char* e = getenv("HELIO_BWD_KSX"); return e ? atoi(e) : 0; }();     // tuning runs
        const int ks_exp = by_rule ? ks_exp_env : 0;
        const bool v3 = variant >= 9;
        CullBwd cull{};
        if (scratch && cull_bwd_possible(variant, B, N, R) && scratch_bytes >= cull_bwd_bytes(B, N))
            cull = launch_cull_bwd(B, N, R, splat_bwd_blocks(R), R, 1, /*with_map=*/false, false, rays, xs, ys, moments, scratch, st);
        // tools/sweep_bwd_nrb.py: from N = 600 — with workgroups enough — no split of the contracted axis at all
        // (4 independent waves of 64 rays each: one epilogue per 2·R MFMAs): B = 25: N = 1000, R = 128: 37 → 30 µs,
        // R = 256: 112 → 98 µs; B = 256, N = 1000, R = 64: 96 → 62 µs; at N = 300 it is 1.5× slower.  (Held to 128
        // registers — four waves per SIMD instead of two — it is no faster: 35.6 µs; the two passes alone take
        // 16.5 and 21.0 µs of the 31 µs they take together.)
        const bool whole_k = variant == 9;
        if (whole_k && cull.counts)
            hipLaunchKernelGGL((splat_bwd_mfma_small<1, 2, 4, true>), dim3(ct * ((N + 255) / 256), B, 2), dim3(256), 0, st, N, R, rays, xs, ys, gimg, moments, cull.counts, cull.idx);
        else if (whole_k || (v3 && ks_exp == 1 && nrb == 2))
            hipLaunchKernelGGL((splat_bwd_mfma_small<1, 2, 4>), dim3(ct * ((N + 255) / 256), B, 2), dim3(256), 0, st, N, R, rays, xs, ys, gimg, moments, nullptr, nullptr);
        else if (v3 && ks_exp == 1 && nrb == 1)
            hipLaunchKernelGGL((splat_bwd_mfma_small<1, 1, 4>), dim3(ct * ((N + 127) / 128), B, 2), dim3(256), 0, st, N, R, rays, xs, ys, gimg, moments, nullptr, nullptr);
        else if (v3 && ks_exp == 2 && nrb == 1)
            hipLaunchKernelGGL((splat_bwd_mfma_small<2, 1, 2>), dim3(ct * ((N + 63) / 64), B, 2), dim3(256), 0, st, N, R, rays, xs, ys, gimg, moments, nullptr, nullptr);
        else if (v3 && ks_exp == 2 && nrb == 2)
            hipLaunchKernelGGL((splat_bwd_mfma_small<2, 2, 2>), dim3(ct * ((N + 127) / 128), B, 2), dim3(256), 0, st, N, R, rays, xs, ys, gimg, moments, nullptr, nullptr);
        else if (nrb == 4)
            hipLaunchKernelGGL((splat_bwd_mfma_small<4, 4>), dim3(ct * ((N + 127) / 128), B, 2), dim3(256), 0, st, N, R, rays, xs, ys, gimg, moments, nullptr, nullptr);
        else if (nrb == 2)
            hipLaunchKernelGGL((splat_bwd_mfma_small<4, 2>), dim3(ct * ((N + 63) / 64), B, 2), dim3(256), 0, st, N, R, rays, xs, ys, gimg, moments, nullptr, nullptr);
        else if (ks == 8)
            hipLaunchKernelGGL(splat_bwd_mfma_small<8>, dim3(ct * nt, B, 2), dim3(512), 0, st, N, R, rays, xs, ys, gimg, moments, nullptr, nullptr);
        else
            hipLaunchKernelGGL(splat_bwd_mfma_small<4>, dim3(ct * nt, B, 2), dim3(256), 0, st, N, R, rays, xs, ys, gimg, moments, nullptr, nullptr);
        return HELIO_OK;
    }
    if (variant == 12) {       // (round 4) the LDS-tile kernel in 64-ray tiles (256- or 128-wide c tiles); lists as variant 2's, their map in 64-ray tiles
        CullBwd cull{};
        if (scratch && cull_bwd_possible(12, B, N, R)) {
            int ct = cull_bwd_ct(12, B, N, R);
            if (scratch_bytes < cull_bwd_bytes(B, N, ct, 64)) ct = 1;
            if (scratch_bytes >= cull_bwd_bytes(B, N, ct, 64))
                cull = launch_cull_bwd(B, N, R, splat_bwd_blocks(R), R <= 128 ? 128 : 256, ct, /*with_map=*/true, false, rays, xs, ys, moments, scratch, st, 64);
        }
        const bool vec = (R & 3) == 0;
        if (R <= 128) {
            if (vec) launch_bwd_mfma_both_v<true, 2, 1>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
            else launch_bwd_mfma_both_v<false, 2, 1>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        } else if (vec) launch_bwd_mfma_both_v<true, 4, 1>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        else launch_bwd_mfma_both_v<false, 4, 1>(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        return HELIO_OK;
    }
    if (variant == 2) {
        CullBwd cull{};
        if (scratch && cull_bwd_possible(2, B, N, R)) {
            // a list per (pass, c tile) where the image is several tiles wide and the scratch holds them, else one
            // per image (what a caller sized by an older query hands over) — the moments are the same bits
            int ct = cull_bwd_ct(2, B, N, R);
            if (scratch_bytes < cull_bwd_bytes(B, N, ct)) ct = 1;
            if (scratch_bytes >= cull_bwd_bytes(B, N, ct))
                cull = launch_cull_bwd(B, N, R, splat_bwd_blocks(R), R <= 128 ? 128 : 256, ct, /*with_map=*/true, bwd_split_tails(R, 2l * B * ct), rays, xs, ys,
                                       moments, scratch, st);
        }
        launch_bwd_mfma_both(B, N, R, rays, xs, ys, gimg, moments, st, cull);
        return HELIO_OK;
    }
    if (variant != 1) return HELIO_E_INVALID;
    const int JB = splat_bwd_blocks(R), NB = (N + BW_NT - 1) / BW_NT;
    hipLaunchKernelGGL(splat_bwd_valu, dim3(JB * NB, B), dim3(256), 0, st, B, N, R, rays, xs, ys, gimg, moments);
    return HELIO_OK;
}

}  // namespace helio
