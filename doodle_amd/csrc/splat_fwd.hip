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
// Three kernels compute that sum:
//   * splat_fwd_valu — the factors of a chunk of rays are staged in LDS, every
//     thread keeps an 8×8 (or 4×4) register tile of pixels and does FMAs.
//   * splat_fwd_mfma_regs — the outer-product sum is issued on the matrix pipe with
//     v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: an exact fmaf chain, same
//     numerics as the VALU kernel); every lane computes its own A/E operand in
//     registers.  Used for small problems (64×64 tiles fill the chip sooner).
//   * splat_fwd_mfma_tile — the throughput kernel: every factor of the workgroup's
//     tile is computed ONCE into LDS in MFMA-operand order and the MFMA loop only reads
//     operands.  Why: measured on MI355X, the f32 MFMA and the VALU do not overlap on
//     a SIMD (f32 MFMA runs at the f32 VALU rate — same multiply-add lanes), so every
//     VALU instruction removed per MFMA is time gained.
// Measured at N=2000, B=512, R=512 (config 4): valu 96, mfma_regs 111, mfma_tile<2> 118,
// mfma_tile<4> 135 TFLOP/s (f32 peak 157.3).
// The 64²-tile kernels (small problems, short sums) accumulate in two levels (a chunk of rays,
// then the running total); the throughput kernels in one level — fewer registers, more waves per
// SIMD — measured 1.5e-6 of peak apart at N=2000 and inside the tolerance at N=5000.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include "helio.h"
#include "ray_trace.h"
#include "step_loss_math.h"
#include "cull.h"

namespace helio {

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// In-kernel stamps of the fused small-problem kernel: ONLY in the diagnostic library
// (tools/build_diag.py compiles this file with -DHELIO_STAMPS into libhelio_diag.so, never into
// libhelio.so).  A stamp is s_memtime (shader cycles) kept in scalar registers; lane 0 of every wave
// writes its stamps once, at the end, to a buffer nothing else reads (cdna_hip_programming.md §7).
#ifdef HELIO_STAMPS
__device__ unsigned long long* g_stamps = nullptr;        // [workgroup][wave][HELIO_NSTAMP]
#define HELIO_NSTAMP 12
#define HSTAMP_DECL unsigned long long stamp_[HELIO_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long stamp_slot_ = 0
// two consecutive launches keep their stamps apart: launch_fused puts the slot helio_diag_set_slot() chose into bit 17 of
// the R argument, and slot 1 writes behind slot 0's [workgroups][waves][HELIO_NSTAMP] block
__device__ long g_stamp_slot_words = 0;
#define HSTAMP_SLOT(x) stamp_slot_ = (long)(x) * g_stamp_slot_words
#define HSTAMP(k)                                                                                   \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_[k]) :: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#define HSTAMP_VM(k)      /* the same after every outstanding vector-memory operation of the wave */ \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_[k]) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#define HSTAMP_REAL(k)                                                                              \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_[k]) :: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#define HSTAMP_FLUSH()                                                                              \
    do {                                                                                            \
        if (g_stamps && (threadIdx.x & 63) == 0) {                                                  \
            unsigned long long* o_ = g_stamps + stamp_slot_ + ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * HELIO_NSTAMP; \
            for (int k_ = 0; k_ < HELIO_NSTAMP; ++k_) o_[k_] = stamp_[k_];                          \
        }                                                                                           \
    } while (0)
#else
#define HSTAMP_DECL
#define HSTAMP_SLOT(x)
#define HSTAMP(k)
#define HSTAMP_VM(k)
#define HSTAMP_REAL(k)
#define HSTAMP_FLUSH()
#endif

// ----------------------------------------------------------------------------------------------
// VALU variant
// ----------------------------------------------------------------------------------------------
template <int TILE, int NC, bool TWO_LEVEL = true>
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
        if (TWO_LEVEL || n0 == 0) {
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int c = 0; c < TC; ++c) acc[r][c] = 0.0f;
        }
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
        if (TWO_LEVEL) {
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int c = 0; c < TC; ++c) tot[r][c] += acc[r][c];
        }
    }
    if (!TWO_LEVEL) {
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int c = 0; c < TC; ++c) tot[r][c] = acc[r][c];
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
// f32 MFMA kernels
// ----------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) const float lds_cf;

// MFMA operand maps (cdna_hip_programming.md §3): lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; the two k of one instruction are two consecutive heliostats.
// C/D map: column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
__device__ __forceinline__ void store_block(float* __restrict__ img, int R, int i0, int j0, int lr, int lh,
                                            const f32x16& v) {
    const int j = j0 + lr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (i < R && j < R) img[(long)i * R + j] = v[e];
    }
}

// Operands in registers.  Workgroup = WI×WJ = 4 waves, wave tile = 32·MBI × 32·MBJ; two
// k-pairs (4 heliostats) per loop trip with the ray fetch one trip ahead; the LDS chunk is
// padded with rays whose A factor is exactly 0, so the loop has no tail logic.
template <int MBI, int MBJ, int WI, int WJ, int NC, bool TWO_LEVEL>
__global__ void __launch_bounds__(256)
splat_fwd_mfma_regs(int B, int Nall, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                    const float* __restrict__ ys, float* __restrict__ image, const int* __restrict__ live_counts,
                    const int* __restrict__ live_order, const float4* __restrict__ live_lists) {
    static_assert(WI * WJ == 4 && NC % 4 == 0, "4 waves per workgroup");
    constexpr int TI = 32 * MBI * WI, TJ = 32 * MBJ * WJ;
    __shared__ float4 sRay[NC + 4];

    const int tiles_j = (R + TJ - 1) / TJ;
    // the workgroup's (image, tile): its place in the grid, or — with lists (cull.h) — the entry of the work
    // order at its linear id (longest list first)
    int b = blockIdx.y, tile = blockIdx.x;
    long list = 0;
    if (live_counts) {
        list = blockIdx.x + gridDim.x * blockIdx.y;
        if (live_order) list = live_order[list];             // (no table where every workgroup starts at once)
        b = (int)(list / gridDim.x);
        tile = (int)(list % gridDim.x);
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int i0 = (tile / tiles_j) * TI + (wave / WJ) * 32 * MBI;
    const int j0 = (tile % tiles_j) * TJ + (wave % WJ) * 32 * MBJ;

    float xv[MBI], yv[MBJ];
#pragma unroll
    for (int m = 0; m < MBI; ++m) xv[m] = xs[min(i0 + 32 * m + lr, R - 1)];
#pragma unroll
    for (int m = 0; m < MBJ; ++m) yv[m] = ys[min(j0 + 32 * m + lr, R - 1)];

    f32x16 tot[MBI][MBJ], acc[MBI][MBJ];
#pragma unroll
    for (int mi = 0; mi < MBI; ++mi)
#pragma unroll
        for (int mj = 0; mj < MBJ; ++mj)
#pragma unroll
            for (int e = 0; e < 16; ++e) { tot[mi][mj][e] = 0.0f; acc[mi][mj][e] = 0.0f; }

    // Rays are staged in LDS pre-scaled, (a·√k2, b·√k2, √k2, c2·k2), so that a factor is
    //   exp2(-(q² + cc)),  q = fma(coord, √k2, shift)        — 2 VALU + 1 exp per factor.
    // A padded ray: √k2 = 0, cc = 1e30  →  A = exp2(-1e30) = 0 exactly, E = 1
    const float4 pad = make_float4(0.f, 0.f, 0.f, 1e30f);
    // the tile's rays: all of the image's, or (cull.h) the ordered list of those that are not exactly zero here
    const float4* __restrict__ rb = reinterpret_cast<const float4*>(rays) + (long)b * Nall;
    int N = Nall;
    if (live_counts) {
        N = live_counts[list];
        rb = live_lists + list * Nall;
    }
    for (int n0 = 0; n0 < N; n0 += NC) {
        __syncthreads();
        for (int k = tid; k < NC + 4; k += 256) {
            float4 v = pad;
            if (k < NC && n0 + k < N) {
                const float4 r = rb[n0 + k];
                const float sk = __builtin_sqrtf(r.z);
                v = make_float4(r.x * sk, r.y * sk, sk, r.w * r.z);
            }
            sRay[k] = v;
        }
        __syncthreads();
        const int cnt = min(NC, N - n0);
        float4 q0 = sRay[lh], q1 = sRay[2 + lh];
        for (int k = 0; k < cnt; k += 4) {
            const float4 p0 = q0, p1 = q1;
            q0 = sRay[k + 4 + lh];
            q1 = sRay[k + 6 + lh];
            float fa0[MBI], fe0[MBJ], fa1[MBI], fe1[MBJ];
            // the two k-pairs' factors as float PAIRS: the fused multiply-adds of a pair are one v_pk_fma_f32 (the same IEEE
            // fma per component — bit-identical factors — in half the issue slots: the f32 MFMA and the VALU share a SIMD's
            // issue, every vector instruction of this loop is MFMA time)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 pz = {p0.z, p1.z}, px = {p0.x, p1.x}, py = {p0.y, p1.y}, pw = {p0.w, p1.w};
#pragma unroll
            for (int m = 0; m < MBI; ++m) {
                const f32x2 c = {xv[m], xv[m]};
                const f32x2 t = __builtin_elementwise_fma(c, pz, px);
                const f32x2 a = __builtin_elementwise_fma(t, t, pw);
                fa0[m] = exp2_fast(-a.x);
                fa1[m] = exp2_fast(-a.y);
            }
#pragma unroll
            for (int m = 0; m < MBJ; ++m) {
                const f32x2 c = {yv[m], yv[m]};
                const f32x2 u = __builtin_elementwise_fma(c, pz, py);
                const f32x2 uu = u * u;
                fe0[m] = exp2_fast(-uu.x);
                fe1[m] = exp2_fast(-uu.y);
            }
#pragma unroll
            for (int mi = 0; mi < MBI; ++mi)
#pragma unroll
                for (int mj = 0; mj < MBJ; ++mj)
                    acc[mi][mj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[mi], fe0[mj], acc[mi][mj], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MBI; ++mi)
#pragma unroll
                for (int mj = 0; mj < MBJ; ++mj)
                    acc[mi][mj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[mi], fe1[mj], acc[mi][mj], 0, 0, 0);
        }
        if (TWO_LEVEL) {
#pragma unroll
            for (int mi = 0; mi < MBI; ++mi)
#pragma unroll
                for (int mj = 0; mj < MBJ; ++mj) {
                    tot[mi][mj] += acc[mi][mj];
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mi][mj][e] = 0.0f;
                }
        }
    }

    float* __restrict__ img = image + (long)b * R * R;
#pragma unroll
    for (int mi = 0; mi < MBI; ++mi)
#pragma unroll
        for (int mj = 0; mj < MBJ; ++mj)
            store_block(img, R, i0 + 32 * mi, j0 + 32 * mj, lr, lh, TWO_LEVEL ? tot[mi][mj] : acc[mi][mj]);
}

// Few images, many heliostats (one sun over a whole plant: B·(R/64)² workgroups of the kernel above do not
// fill the chip, and each walks all N rays alone — 30 ns a ray, 150 µs at N = 5000 whatever B is).  Here a
// workgroup owns ONE 32×32 block and its KP waves SPLIT THE HELIOSTAT SUM.  The sum is always cut into
// KSPLIT_PARTS = 16 parts of `per` consecutive rays, each accumulated from zero; a wave takes 16 / KP of
// them (one accumulator each), and the 16 partial blocks meet in LDS at the end and are added in part
// order — so the bits do not depend on KP, i.e. not on how many images the call renders (a shard of a
// batch gives the rows the whole batch would: doodle_amd/sharded.py).  A part's rays are staged 64 at a
// time — one ray per lane, pre-scaled as above — in a buffer of LDS private to the wave, double-buffered
// (the next 64 are requested before the current 64 are consumed; a wave's LDS operations are ordered, so
// there is no workgroup barrier in the loop).  B·(R/32)² workgroups × KP waves: 16× the waves of the kernel
// above at KP = 16.
constexpr int KSPLIT_PARTS = 16;

template <int KP>
__global__ void __launch_bounds__(64 * KP)
splat_fwd_block_ksplit(int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                       const float* __restrict__ ys, float* __restrict__ image, const int* __restrict__ live_counts,
                       const float4* __restrict__ live_lists) {
    constexpr int PW = KSPLIT_PARTS / KP;              // parts per wave
    // ray buffers [KP][2][64 + 4] float4 during the sum; afterwards the same space holds the partial blocks,
    // 8 of the 16 accumulator registers at a time: [16 parts][8][64] floats (35 KB at KP = 16)
    constexpr int RAY_FLOATS = KP * 2 * 68 * 4, RED_FLOATS = KSPLIT_PARTS * 8 * 64;
    __shared__ __attribute__((aligned(16))) float smem[RAY_FLOATS > RED_FLOATS ? RAY_FLOATS : RED_FLOATS];
    const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const float xv = xs[min(i0 + lr, R - 1)], yv = ys[min(j0 + lr, R - 1)];

    const int per = (((N + KSPLIT_PARTS - 1) / KSPLIT_PARTS) + 3) & ~3;
    const float4* __restrict__ rb = reinterpret_cast<const float4*>(rays) + (long)b * N;

    f32x16 acc[PW];
    int buf = 0;
#pragma unroll
    for (int pw = 0; pw < PW; ++pw) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pw][e] = 0.0f;
        const int part = wave * PW + pw;
        // the part's rays: `per` consecutive ones of the image, or (cull.h) the ordered list of those of them whose
        // footprint is not exactly zero on the image — each part is a chain from zero either way: the same bits
        int n_begin = min(N, per * part), n_end = min(N, n_begin + per), last = N - 1;
        const float4* __restrict__ src = rb;
        if (live_counts) {
            const long list = (long)b * KSPLIT_PARTS + part;
            n_begin = 0;
            n_end = live_counts[list];
            src = live_lists + list * per;
            last = max(n_end - 1, 0);
        }
        // (clamped, unconditional loads: "in range ? load : pad" becomes a branch around the load)
        auto fetch = [&](int base) { return src[min(base + lane, last)]; };
        float4 nxt = fetch(n_begin);
        for (int base = n_begin; base < n_end; base += 64, buf ^= 1) {
            float4* __restrict__ tab = reinterpret_cast<float4*>(smem) + (wave * 2 + buf) * 68;
            {
                const float sk = __builtin_sqrtf(nxt.z);
                // a ray past this part: √k2 = 0, cc = 1e30 → A = exp2(-1e30) = 0 exactly, E = 1
                tab[lane] = base + lane < n_end ? make_float4(nxt.x * sk, nxt.y * sk, sk, nxt.w * nxt.z)
                                                : make_float4(0.f, 0.f, 0.f, 1e30f);
            }
            if (lane < 4) tab[64 + lane] = make_float4(0.f, 0.f, 0.f, 1e30f);   // what the look-ahead reads past the end
            if (base + 64 < n_end) nxt = fetch(base + 64);                       // in flight during this chunk
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int cnt = min(64, n_end - base);
            float4 q0 = tab[lh], q1 = tab[2 + lh];
            for (int k = 0; k < cnt; k += 4) {
                const float4 p0 = q0, p1 = q1;
                q0 = tab[k + 4 + lh];
                q1 = tab[k + 6 + lh];
                const float t0 = __builtin_fmaf(xv, p0.z, p0.x), t1 = __builtin_fmaf(xv, p1.z, p1.x);
                const float u0 = __builtin_fmaf(yv, p0.z, p0.y), u1 = __builtin_fmaf(yv, p1.z, p1.y);
                const float fa0 = exp2_fast(-__builtin_fmaf(t0, t0, p0.w)), fa1 = exp2_fast(-__builtin_fmaf(t1, t1, p1.w));
                const float fe0 = exp2_fast(-(u0 * u0)), fe1 = exp2_fast(-(u1 * u1));
                acc[pw] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fe0, acc[pw], 0, 0, 0);
                acc[pw] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fe1, acc[pw], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();   // the other buffer is rewritten only after every lane has left this chunk
        }
    }

    // element (e, l) of the block: row (e&3) + 8(e>>2) + 4(l>>5), column l&31 (the MFMA's C/D map); thread t
    // adds the 16 partials of elements t, t + 64·KP, … in part order — lanes ↔ consecutive columns
    float* __restrict__ img = image + (long)b * R * R;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                       // every wave has left its ray buffers / the previous half is summed
#pragma unroll
        for (int pw = 0; pw < PW; ++pw)
#pragma unroll
            for (int e = 0; e < 8; ++e) smem[((wave * PW + pw) * 8 + e) * 64 + lane] = acc[pw][8 * half + e];
        __syncthreads();
        for (int idx = tid; idx < 512; idx += 64 * KP) {
            const int e8 = idx >> 6, l = idx & 63, e = 8 * half + e8;
            float v = smem[e8 * 64 + l];
#pragma unroll
            for (int p = 1; p < KSPLIT_PARTS; ++p) v += smem[(p * 8 + e8) * 64 + l];
            const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), j = j0 + (l & 31);
            if (i < R && j < R) img[(long)i * R + j] = v;
        }
    }
}

// the shapes it serves: more heliostats than the single-launch kernel's table holds, and so few 64×64 tiles
// that the register-operand kernel would leave most of the chip idle; → waves per workgroup (0: not this kernel)
static int ksplit_parts(int B, int N, int R) {
    static const int forced = [] { const char* e = getenv("HELIO_KSPLIT"); return e ? atoi(e) : -1; }();
    if (forced == 0) return 0;
    const long blocks = (long)B * ((R + 31) / 32) * ((R + 31) / 32);
    if (N < 128) return 0;
    if (forced == 4 || forced == 8 || forced == 16) return forced;      // tuning runs
    // tools/sweep_ksplit.py, tools/sweep_ksplit_kp.py: 16 waves a block are the best or within 3 % of it at
    // every block count it was measured at (4 and 8 waves: 3–15 % behind); against the 64² and 128² tile
    // kernels it wins up to 1024 blocks from N = 128 and up to 2048 blocks from N = 500 (5–25 %; at N = 300
    // and 2048 blocks the 64² kernel is 10 % ahead: the LDS reduce), and loses to the 128² kernel at 4096
    if (blocks <= 1024) return N >= 256 ? 16 : 8;
    return (blocks <= 2048 && N >= 500) ? 16 : 0;
}

static bool launch_ksplit(int B, int N, int R, const float* rays, const float* xs, const float* ys, float* image,
                          int kp, hipStream_t st, CullFwd cull = CullFwd{nullptr, nullptr, nullptr}) {
    const int t = (R + 31) / 32;
    if (t > 65535 || B > 65535) return false;
    const dim3 grid(t, t, B);
    if (kp == 16) hipLaunchKernelGGL(splat_fwd_block_ksplit<16>, grid, dim3(1024), 0, st, N, R, rays, xs, ys, image, cull.counts, cull.lists);
    else if (kp == 8) hipLaunchKernelGGL(splat_fwd_block_ksplit<8>, grid, dim3(512), 0, st, N, R, rays, xs, ys, image, cull.counts, cull.lists);
    else if (kp == 4) hipLaunchKernelGGL(splat_fwd_block_ksplit<4>, grid, dim3(256), 0, st, N, R, rays, xs, ys, image, cull.counts, cull.lists);
    else return false;
    return true;
}

// Operands through LDS.  Workgroup = W×W waves of 64×64 pixels (W = 2: 256 threads, 128×128
// tile, two-level sums, two workgroups per CU; W = 4: 1024 threads, 256×256 tile, one level,
// one workgroup per CU).  Per chunk of 64 rays:
//   producer phase — thread `lane` of every wave owns ray `lane` of the chunk (parameters in
//     registers, prefetched one chunk ahead) and fills 32-pixel groups of the factor tables,
//     pixel index wave-uniform (coordinates broadcast from LDS):
//       factor = exp2(-(q² + cc)),  q = fma(coord, sk, shift·sk),  sk = sqrt(k2)   (2 VALU + 1 exp)
//     Table rows are padded to an odd length, so the per-ray writes (lanes ↔ rays) and the
//     per-pixel operand reads (lanes ↔ rows) are both bank-conflict free.
//   consumer phase — 32 k-pairs × 4 MFMAs per wave, fully unrolled; operands are ds_read_b32
//     from four base registers + 16-bit immediates (no VALU instruction in the loop).
#ifndef HELIO_FWD_LAST_GROUP
#define HELIO_FWD_LAST_GROUP 16        // rays per group of the consumer loop (A/B builds: 8, 16, 64 = no groups)
#endif
template <int W, bool TWO_LEVEL>
__global__ void __launch_bounds__(64 * W * W)
splat_fwd_mfma_tile(int B, int Nall, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                    const float* __restrict__ ys, float* __restrict__ image, const int* __restrict__ live_counts,
                    const int* __restrict__ live_order, const float4* __restrict__ live_lists, int S, int P,
                    float* __restrict__ partials) {
    // row pitch: odd for the 4-wave form (ds_write_b32 per factor), T+2 for the 16-wave form, whose
    // producer stores factor PAIRS with ds_write_b64 (8-byte aligned rows; 16 lanes × 2 dwords
    // at pitch 258 cover the 32 banks exactly once)
    constexpr int NC = 64, T = 64 * W, LD = T + (W == 4 ? 2 : 1), NW = W * W;
    constexpr int GROUPS = 2 * T / 32 / NW;          // 32-pixel factor groups per wave (1 or 2)
    static_assert(GROUPS * NW * 32 == 2 * T, "groups must tile the two tables");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // sA[NC][LD] sE[NC][LD] xs[T] ys[T]
    float* __restrict__ sXY = smem + 2 * NC * LD;

    const int tiles_j = (R + T - 1) / T;
    // the workgroup's (image, tile): its place in the grid, or — with lists (cull.h) — the entry of the work
    // order at its linear id (longest list first)
    // With S > 1 the heliostat sum is SPLIT: blockIdx.x = tile·S + part, part p sums the rays [p·P, p·P + P) into
    // partials[b][p] (an R×R image of its own), which splat_reduce_parts adds in part order (launch_splat_fwd).
    int b = blockIdx.y, tile = blockIdx.x;
    long list = 0;
    if (live_counts) {
        list = blockIdx.x + gridDim.x * blockIdx.y;
        if (live_order) list = live_order[list];             // (no table where every workgroup starts at once)
        b = (int)(list / gridDim.x);
        tile = (int)(list % gridDim.x);
    }
    int part = 0;
    if (S > 1) { part = tile % S; tile /= S; }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int ti0 = (tile / tiles_j) * T, tj0 = (tile % tiles_j) * T;
    const int wi = (wave / W) * 64, wj = (wave % W) * 64;

    for (int k = tid; k < 2 * T; k += 64 * NW)
        sXY[k] = k < T ? xs[min(ti0 + k, R - 1)] : ys[min(tj0 + k - T, R - 1)];

    f32x16 tot[4], acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { tot[m][e] = 0.0f; acc[m][e] = 0.0f; }

    const float4 pad = make_float4(0.f, 0.f, 1.f, 1e30f);   // A = exp2(-1e30) = 0 exactly
    // the tile's rays: all of the image's, or (cull.h) the ordered list of those that are not exactly zero here
    const float4* __restrict__ rb = reinterpret_cast<const float4*>(rays) + (long)b * Nall;
    int N = Nall;
    if (S > 1) { rb += part * P; N = max(0, min(Nall - part * P, P)); }
    if (live_counts) {
        N = live_counts[list];
        rb = live_lists + list * (S > 1 ? P : Nall);
    }
    float4 g = (lane < N) ? rb[lane] : pad;

    lds_cf* pa = (lds_cf*)smem + lh * LD + wi + lr;
    lds_cf* pe = (lds_cf*)smem + NC * LD + lh * LD + wj + lr;
    lds_cf* pa1 = pa + 32;
    lds_cf* pe1 = pe + 32;
    // four independent base registers: a merged ds_read2_b32 has 8-bit offsets and one
    // shared base cannot reach the second (>64 KB away) table with a 16-bit immediate
    asm volatile("" : "+v"(pa));
    asm volatile("" : "+v"(pe));
    asm volatile("" : "+v"(pa1));
    asm volatile("" : "+v"(pe1));

    // 16-wave form: a wave's 32 pixel coordinates never change and are wave-uniform — keep them
    // in scalar registers for the whole kernel (s_load; no LDS read in the producer loop)
    float cs[32];
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    if constexpr (GROUPS == 1) {
        const bool sa = uw < T / 32;
        const float* __restrict__ cptr = sa ? xs : ys;
        const int cbase = (sa ? ti0 : tj0) + (sa ? uw : uw - T / 32) * 32;
#pragma unroll
        for (int j = 0; j < 32; ++j) cs[j] = cptr[min(cbase + j, R - 1)];
    }

    for (int n0 = 0; n0 < N; n0 += NC) {
        const float4 q = g;
        const float sk = __builtin_sqrtf(q.z);
        __syncthreads();                                     // tables free (and sXY visible)
        if (n0 + NC < N) g = (n0 + NC + lane < N) ? rb[n0 + NC + lane] : pad;   // in flight during the chunk
        if constexpr (GROUPS == 1) {
            const bool is_a = uw < T / 32;
            const int p0 = (is_a ? uw : uw - T / 32) * 32;
            const float shift = (is_a ? q.x : q.y) * sk;
            const float cc = is_a ? q.w * q.z : 0.0f;
            lds_f* wdst = (lds_f*)smem + (is_a ? 0 : NC * LD) + lane * LD + p0;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef __attribute__((address_space(3))) f32x2 lds_f2;
            // two pixels per instruction: the two fused multiply-adds of a factor pair are v_pk_fma_f32
            // (the same IEEE fma per component — bit-identical factors — in half the issue slots; the
            // f32 MFMA and the VALU share a SIMD's issue, so producer instructions are MFMA time)
            const f32x2 sk2 = {sk, sk}, shift2 = {shift, shift}, cc2 = {cc, cc};
#pragma unroll
            for (int j = 0; j < 32; j += 2) {
                const f32x2 c2 = {cs[j], cs[j + 1]};
                const f32x2 t = __builtin_elementwise_fma(c2, sk2, shift2);
                const f32x2 a = __builtin_elementwise_fma(t, t, cc2);
                f32x2 v;
                v.x = exp2_fast(-a.x);
                v.y = exp2_fast(-a.y);
                *reinterpret_cast<lds_f2*>(wdst + j) = v;
            }
        } else
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi) {
            const int grp = wave * GROUPS + gi;              // wave-uniform: [0, T/32) → A, rest → E
            const bool is_a = grp < T / 32;
            const int p0 = (is_a ? grp : grp - T / 32) * 32;
            const float shift = (is_a ? q.x : q.y) * sk;
            const float cc = is_a ? q.w * q.z : 0.0f;
            lds_f* wdst = (lds_f*)smem + (is_a ? 0 : NC * LD) + lane * LD + p0;
            const float* __restrict__ coord = sXY + (is_a ? 0 : T) + p0;
#pragma unroll
            for (int j4 = 0; j4 < 32; j4 += 4) {
                const float4 cv = *reinterpret_cast<const float4*>(&coord[j4]);
                const float ca[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = __builtin_fmaf(ca[j], sk, shift);
                    wdst[j4 + j] = exp2_fast(-__builtin_fmaf(t, t, cc));
                }
            }
        }
        __syncthreads();
        // the short last chunk: its padded rays have A = 0 exactly and add +0 to sums that are never negative, so the
        // k-pairs made of padding alone are left out — GRP rays at a time, a scalar compare and branch between the groups
        // of the unrolled chunk — and the bits stay.  Measured (tools/ab_fwd_last.sh, profiles/r04_m_fwd_last_chunk.txt; GRP =
        // 64, i.e. every chunk whole as before round 4 / 8 / 16): config 4 (N = 2000: 16 rays in its last chunk) dense
        // 3926 / 3857 / 3826 µs, with the lists (half a chunk saved per list) 1986 / 1941 / 1938; B = 256, N = 200, R = 512:
        // 281 / 236 / 241; B = 500, N = 200, R = 256: 147 / 126 / 126.  The groups cost a whole chunk nothing (the operand
        // reads of a group are issued together at its head; the other waves of the SIMD cover them).
        constexpr int GRP = HELIO_FWD_LAST_GROUP;
        const int steps = __builtin_amdgcn_readfirstlane(min(NC / GRP, (N - n0 + GRP - 1) / GRP));
#pragma unroll
        for (int s8 = 0; s8 < NC / GRP; ++s8) {
            if (s8 > 0 && s8 >= steps) break;
#pragma unroll
            for (int kp = GRP / 2 * s8; kp < GRP / 2 * (s8 + 1); ++kp) {
                const float a0 = pa[kp * 2 * LD], a1 = pa1[kp * 2 * LD];
                const float e0 = pe[kp * 2 * LD], e1 = pe1[kp * 2 * LD];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, e0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, e1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, e0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, e1, acc[3], 0, 0, 0);
            }
        }
        if (TWO_LEVEL) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                tot[m] += acc[m];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;
            }
        }
    }

    float* __restrict__ img = S > 1 ? partials + ((long)b * S + part) * R * R : image + (long)b * R * R;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        store_block(img, R, ti0 + wi + 32 * (m >> 1), tj0 + wj + 32 * (m & 1), lr, lh, TWO_LEVEL ? tot[m] : acc[m]);
}

// image[b] = ((partials[b][0] + partials[b][1]) + partials[b][2]) + … — the parts of a split heliostat sum, added
// in part order (round to nearest each): the bits depend on N, R and S, not on which workgroup ran when
__global__ void __launch_bounds__(256)
splat_reduce_parts(long pixels_per_image, int S, const float* __restrict__ partials, float* __restrict__ image) {
    const int b = blockIdx.y;
    const float* __restrict__ src = partials + (long)b * S * pixels_per_image;
    float* __restrict__ dst = image + (long)b * pixels_per_image;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < pixels_per_image; p += (long)gridDim.x * blockDim.x) {
        float v = src[p];
        for (int k = 1; k < S; ++k) v += src[(long)k * pixels_per_image + p];
        dst[p] = v;
    }
}

// Small problems are launch-latency bound (config 2 is 41 MFLOP ≈ 0.3 µs at peak): ONE launch for
// the whole forward, shaped for the latency of a single wave rather than for throughput.  Stamps of
// the round-1 form (64×64 tiles, 4 waves, one wave tracing every ray of the sun; profiles/
// r02_a_fused_stamps_before.txt) showed a 9,700-cycle wave of which 3,800 were the heliostat loop,
// 2,000 the trace and 1,300 sixteen bounds-checked stores per lane — every phase serial instruction
// issue of one wave per SIMD, with 100 of the 256 CUs in use.  So here:
//   * a workgroup is ONE 32×32 pixel block (one MFMA accumulator) and KG waves that split the
//     heliostats between them (KG ∈ {1,2,4}: config 2 runs 400 workgroups of 2 or 4 waves);
//   * the rays of the sun are traced by ONE wave per 64 rays (lane ↔ ray, the bit-faithful trace() of
//     ray_trace.h) into an LDS table while the other waves wait at the barrier: a first form in which
//     every wave traced the rays of its own k-pairs put 1,600 copies of that 350-instruction stream on
//     1,024 SIMDs, and the workgroups that shared a CU finished 1 µs after the others;
//   * the KG partial accumulators are exchanged through LDS so that every wave ends up OWNING 16/KG
//     registers (8/KG·… rows) of the block, summed in fixed wave order (deterministic), and stores
//     only those: the store phase is split KG ways too;
//   * block 0 of an image also writes `actual`, `refl` and the `rays` work buffer (each wave its rays).
// The geometry is redundant across the blocks of an image (a few hundred flops per ray and block): it
// costs no time, the lanes are there.
//
// LOSS = true is HelioEnv.step's small-problem forward in the same launch (test_environment.py
// :416-457): the block's share of the three image sums is taken from the owned accumulator registers
// (the image is still written — it is an output of step() — but never re-read; the target and
// distance-map pixels are fetched before the heliostat loop), and ONE EXTRA workgroup per image
// (blockIdx.x == blocks) does the per-ray side work: it writes `actual` / `refl` / `rays`, evaluates
// the two ray losses and fills the `aux` row.  Partials go to the [B, chunks, 3] / [ray_wgs, 2] layout
// step_losses_final reduces (chunks = blocks per image, ray_wgs = B); fixed order, no atomics.
// The kernel arguments behind the 14 preloaded dwords.  A freshly launched kernel waits ≈1,900 cycles
// (0.8 µs; stamps, profiles/r02_*) for its FIRST read of the kernel-argument segment, and hipcc puts
// that wait wherever the first late argument is used — in practice in front of the first ray load.
// The forward-only kernel therefore fetches this block ITSELF: three s_load instructions issued as its
// first instructions, ONE wait placed after the half of the trace that needs none of it
// (cdna_hip_programming.md §5.7, form (ii): "=s" loads, then a wait statement naming every destination
// "+s").  No compiler-counted LDS or scalar-memory operation lies between issue and wait, the
// destinations stay in the registers they were loaded into (audit: tools/audit_fused_late.py), and
// the C++ side never names the `late` parameter, so hipcc emits no load of its own for it.
struct FusedLate {
    PlaneK P;                                          // 19 floats
    float* actual; float* refl; float* rays; float* image;
};
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int FUSED_LATE_AT = 56;                      // bytes of N, R|flag and six pointers in front of it
static_assert(sizeof(PlaneK) == 76 && offsetof(FusedLate, actual) == 80 && offsetof(FusedLate, image) == 104 &&
              sizeof(FusedLate) == 112, "the s_load offsets below restate this layout");

struct LateRegs { i32x16 a; i32x8 b; i32x4 c; };       // floats 0..15 of P | P[16..18], pad, actual, refl | rays, image

__device__ __forceinline__ void late_issue(LateRegs& r) {
    auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("s_load_dwordx16 %0, %3, 0x38\n\ts_load_dwordx8 %1, %3, 0x78\n\ts_load_dwordx4 %2, %3, 0x98"
                 : "=&s"(r.a), "=&s"(r.b), "=&s"(r.c) : "s"(kp));      // early-clobber: no destination on top of the base pair
}
__device__ __forceinline__ void late_wait(LateRegs& r) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a), "+s"(r.b), "+s"(r.c) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float* late_ptr(int lo, int hi) {
    return reinterpret_cast<float*>(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ PlaneK late_plane(const LateRegs& r) {
    auto f = [](int v) { return __int_as_float(v); };
    PlaneK P;
    P.o = {f(r.a[0]), f(r.a[1]), f(r.a[2])};
    P.nrm = {f(r.a[3]), f(r.a[4]), f(r.a[5])};
    P.u = {f(r.a[6]), f(r.a[7]), f(r.a[8])};
    P.v = {f(r.a[9]), f(r.a[10]), f(r.a[11])};
    P.w = {f(r.a[12]), f(r.a[13]), f(r.a[14])};
    P.phat = {f(r.a[15]), f(r.b[0]), f(r.b[1])};
    P.sigma_scale = f(r.b[2]);
    return P;
}

template <int KG, bool LOSS>
__global__ void __launch_bounds__(64 * KG)
render_fwd_fused_small(int N, int R_and_flag, const float* __restrict__ helios, const float* __restrict__ sun,
                       const float* __restrict__ action, const float* __restrict__ trig,
                       const float* __restrict__ xs, const float* __restrict__ ys, FusedLate late, StepLossArgs L) {
    // Argument order: the first 14 dwords — N, R (with the per-sun flag of the trig table, stride 4N or 0,
    // in bit 16) and the six pointers the ray loads and the pixel-coordinate loads need — are PRELOADED
    // into scalar registers with the wave launch (-mllvm -amdgpu-kernarg-preload-count=14,
    // doodle_amd/build.py).  N <= 64·KG: one wave-load of rays per wave, no loop around the trace.
    const int R = R_and_flag & 0xFFFF;
    const long trig_b_stride = ((R_and_flag >> 16) & 1) ? 4l * N : 0l;
    constexpr int OWN = 16 / KG;                       // accumulator registers a wave owns after the exchange
    constexpr int Q4 = OWN / 4;                        // … as 16-byte groups
    __shared__ float4 sRay[64 * KG + 4];
    __shared__ float4 sRed[KG > 1 ? KG * KG * Q4 * 64 : 1];
    __shared__ float scratch[3 * KG];

    // grid = (column blocks, row blocks [+ 1 row of ray workgroups with LOSS], suns): the block's place
    // comes with the launch — an integer division by a kernel argument would put a scalar-load round
    // trip and 25 instructions in front of the first ray load
    const int b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    HSTAMP_DECL;
    HSTAMP_SLOT((R_and_flag >> 17) & 1);               // (diagnostic build: which half of the stamp buffer this launch writes)
    HSTAMP_REAL(8);
    HSTAMP(0);
    LateRegs lr_;
    if constexpr (!LOSS) late_issue(lr_);              // first instructions of the kernel; waited for after trace_head
    // the sun through the VECTOR memory path (an opaque zero lane offset): as a scalar load its wait would
    // be an lgkmcnt(0), which also waits for the late arguments just requested
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const vec3 s = ld3(sun + 3l * b + zoff);

    if constexpr (LOSS) {
        if (blockIdx.y == gridDim.y - 1) {               // the extra grid row: ONE per-ray workgroup per image
            if (blockIdx.x != 0) return;
            float sa = 0.0f, sb = 0.0f;
            float* __restrict__ actual = late.actual;
            float* __restrict__ refl = late.refl;
            float* __restrict__ rays = late.rays;
            const int n = tid;
            if (n < N) {
                const long m = (long)b * N + n;
                const float4 tg = *reinterpret_cast<const float4*>(trig + (long)b * trig_b_stride + 4l * n);
                const vec3 v = ld3(action + 3 * m);
                const Ray q = trace(v, tg.x, tg.y, tg.z, tg.w, ld3(helios + 3l * n), s, late.P);
                st3(actual + 3 * m, q.act);
                if (refl) st3(refl + 3 * m, q.r);
                if (rays) *reinterpret_cast<float4*>(rays + 4 * m) = make_float4(q.a, q.b, q.k2, q.c2);
                const float act[3] = {q.act.x, q.act.y, q.act.z};
                const RayLoss r = ray_loss(L.ideal + 3 * m, act, action + 3 * m, helios + 3l * n, L.g);
                L.align_err[m] = r.ang;
                L.all_bounds[m] = r.out;
                if (L.aux) {     // observation row [sun_b, action_b] (test_environment.py:424)
                    float* a = L.aux + (long)b * (3 + 3l * N);
                    a[3 + 3 * n] = v.x; a[4 + 3 * n] = v.y; a[5 + 3 * n] = v.z;
                    if (n == 0) { a[0] = s.x; a[1] = s.y; a[2] = s.z; }
                }
                sa = r.ang;
                sb = L.g.exponential_risk ? expf(r.out + 1e-6f) : r.out;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
            if (lane == 0) { scratch[2 * wave] = sa; scratch[2 * wave + 1] = sb; }
            __syncthreads();
            if (tid == 0) {
                float ta = scratch[0], tb = scratch[1];
#pragma unroll
                for (int w = 1; w < KG; ++w) { ta += scratch[2 * w]; tb += scratch[2 * w + 1]; }
                L.part_ray[2l * b] = ta;
                L.part_ray[2l * b + 1] = tb;
            }
            return;
        }
    }

    const int lr = lane & 31, lh = lane >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const bool writer = !LOSS && blockIdx.x == 0 && blockIdx.y == 0;

    const float xv = xs[min(i0 + lr, R - 1)], yv = ys[min(j0 + lr, R - 1)];
    const int col = j0 + lr;

    // wave w traces rays 64 w + lane (a wave with no ray — waves 1.. at N <= 64 — only fills its part of
    // the table with padding and waits: ONE instruction stream per 64 rays, so that the trace never
    // competes with itself for a SIMD), then the k-pairs are dealt evenly to the KG waves
    const float4 pad = make_float4(0.f, 0.f, 0.f, 1e30f);   // pre-scaled form: A = exp2(-1e30) = 0 exactly
    const int n = 64 * wave + lane;
    const bool tracing = n < N;
    const long m = (long)b * N + n;
    Ray q;
    vec3 hv = {0.f, 0.f, 0.f};
    if (tracing) {
        const float4 tg = *reinterpret_cast<const float4*>(trig + (long)b * trig_b_stride + 4l * n);
        const vec3 av = ld3(action + 3 * m);
        hv = ld3(helios + 3l * n);
        trace_head(q, av, tg.x, tg.y, tg.z, tg.w, hv, s);
    }
    HSTAMP(1);
    // the late arguments: ONE wait, on every path, in front of their first use
    PlaneK P;
    float* __restrict__ actual;
    float* __restrict__ refl;
    float* __restrict__ rays;
    float* __restrict__ img;
    if constexpr (!LOSS) {
        late_wait(lr_);
        P = late_plane(lr_);
        actual = late_ptr(lr_.b[4], lr_.b[5]);
        refl = late_ptr(lr_.b[6], lr_.b[7]);
        rays = late_ptr(lr_.c[0], lr_.c[1]);
        img = late_ptr(lr_.c[2], lr_.c[3]) + (long)b * R * R;
    } else {
        P = late.P;
        actual = late.actual; refl = late.refl; rays = late.rays;
        img = late.image + (long)b * R * R;
    }
    HSTAMP(7);
    // LOSS: this lane's OWN target / distance-map pixels (the rows this wave owns after the exchange:
    // accumulator registers wave·OWN + e, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)), in flight
    // during the rest of the trace and the loop
    float tgt[OWN], dmp[OWN];
    if constexpr (LOSS) {
        const long base = (long)b * R * R;
#pragma unroll
        for (int e = 0; e < OWN; ++e) {
            const int eg = wave * OWN + e;
            const int i = i0 + (eg & 3) + 8 * (eg >> 2) + 4 * lh;
            const bool in = i < R && col < R;
            tgt[e] = in ? L.target[base + (long)i * R + col] : 0.0f;
            dmp[e] = in ? L.dmaps[base + (long)i * R + col] : 0.0f;
        }
    }

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

    {
        float4 v = pad;
        if (tracing) {
            trace_tail(q, hv, P);
            const float sk = __builtin_sqrtf(q.k2);
            v = make_float4(q.a * sk, q.b * sk, sk, q.c2 * q.k2);   // LDS copy pre-scaled (2 VALU + exp per factor)
            if (writer) {
                st3(actual + 3 * m, q.act);
                if (refl) st3(refl + 3 * m, q.r);
                if (rays) *reinterpret_cast<float4*>(rays + 4 * m) = make_float4(q.a, q.b, q.k2, q.c2);
            }
        }
        sRay[64 * wave + lane] = v;
        if (tid < 4) sRay[64 * KG + tid] = pad;
        __syncthreads();
        HSTAMP(2);
        const int cnt = N;                                   // rays in the table (N <= 64·KG)
        const int per = (((cnt + 1) >> 1) + KG - 1) / KG;    // k-pairs per wave
        const int k_begin = min(cnt, 2 * per * wave), k_end = min(cnt, 2 * per * (wave + 1));
        // groups of 8 k-pairs (16 rays): the group's eight operand fetches are issued together, ahead of
        // its MFMAs (left in a rotating loop, hipcc re-forms "read, wait, use" and every trip pays an LDS
        // latency); a uniform test per k-pair ends the group at the last ray
        for (int k0 = k_begin; k0 < k_end; k0 += 16) {
            float4 p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = sRay[min(k0 + 2 * j, 64 * KG) + lh];     // (64·KG …: padding rays)
#pragma unroll
            for (int j = 0; j < 8; ++j)            // pin the fetches here: hipcc otherwise sinks each into its k-pair
                asm volatile("" :: "v"(p[j].x), "v"(p[j].y), "v"(p[j].z), "v"(p[j].w));
            __builtin_amdgcn_sched_barrier(0);
            // the group's 16 factors first (independent VALU work, issued back to back), then its MFMAs
            // back to back: the f32 MFMA holds the SIMD's vector issue for its 64 cycles, so "factor, factor,
            // MFMA" per k-pair pays every exponential's latency in front of every MFMA (190 cycles per k-pair
            // measured, against 64 + 6 issue slots)
            float fa[8], fe[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = __builtin_fmaf(xv, p[j].z, p[j].x), u = __builtin_fmaf(yv, p[j].z, p[j].y);
                fa[j] = exp2_fast(-__builtin_fmaf(t, t, p[j].w));
                fe[j] = exp2_fast(-(u * u));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(fa[j]), "v"(fe[j]));    // (pinned, as the fetches are)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k0 + 2 * j >= k_end) break;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fe[j], acc, 0, 0, 0);
            }
        }
    }
    HSTAMP(3);

    // exchange: wave w writes its partial sums of the registers wave o owns to slot [w][o]; after the
    // barrier wave o adds the KG partials of its registers in wave order
    float own[OWN];
    if constexpr (KG == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) own[e] = acc[e];
    } else {
#pragma unroll
        for (int o = 0; o < KG; ++o)
#pragma unroll
            for (int g4 = 0; g4 < Q4; ++g4)
                sRed[((wave * KG + o) * Q4 + g4) * 64 + lane] =
                    make_float4(acc[o * OWN + 4 * g4], acc[o * OWN + 4 * g4 + 1], acc[o * OWN + 4 * g4 + 2], acc[o * OWN + 4 * g4 + 3]);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < KG; ++w)
#pragma unroll
            for (int g4 = 0; g4 < Q4; ++g4) {
                const float4 v = sRed[((w * KG + wave) * Q4 + g4) * 64 + lane];
                if (w == 0) { own[4 * g4] = v.x; own[4 * g4 + 1] = v.y; own[4 * g4 + 2] = v.z; own[4 * g4 + 3] = v.w; }
                else { own[4 * g4] += v.x; own[4 * g4 + 1] += v.y; own[4 * g4 + 2] += v.z; own[4 * g4 + 3] += v.w; }
            }
    }
    HSTAMP(4);

    const int r0 = i0 + 4 * lh;                                  // + (reg & 3) + 8 (reg >> 2)
    if (i0 + 32 <= R && j0 + 32 <= R) {                          // (uniform) the block is inside the image
#pragma unroll
        for (int e = 0; e < OWN; ++e) {
            const int eg = wave * OWN + e;
            img[(long)(r0 + (eg & 3) + 8 * (eg >> 2)) * R + col] = own[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < OWN; ++e) {
            const int eg = wave * OWN + e;
            const int i = r0 + (eg & 3) + 8 * (eg >> 2);
            if (i < R && col < R) img[(long)i * R + col] = own[e];
        }
    }
    HSTAMP(5);
    HSTAMP_VM(6);
    HSTAMP_REAL(9);
    HSTAMP_FLUSH();
    if constexpr (LOSS) {
        const float sc = L.tx[b];
        float sq = 0.f, ab = 0.f, ds = 0.f;
#pragma unroll
        for (int e = 0; e < OWN; ++e) {
            const int eg = wave * OWN + e;
            const int i = r0 + (eg & 3) + 8 * (eg >> 2);
            if (i < R && col < R) {
                const float d = own[e] / sc - tgt[e] / sc;      // as the reference divides (:438-441)
                const float ad = fabsf(d);
                sq = __builtin_fmaf(d, d, sq);
                ab += ad;
                ds = __builtin_fmaf(ad, dmp[e], ds);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { sq += __shfl_xor(sq, d); ab += __shfl_xor(ab, d); ds += __shfl_xor(ds, d); }
        if (lane == 0) { scratch[3 * wave] = sq; scratch[3 * wave + 1] = ab; scratch[3 * wave + 2] = ds; }
        __syncthreads();
        if (tid < 3) {
            float t = scratch[tid];
#pragma unroll
            for (int w = 1; w < KG; ++w) t += scratch[3 * w + tid];
            L.part_img[3l * ((long)b * (gridDim.x * (gridDim.y - 1)) + blockIdx.y * gridDim.x + blockIdx.x) + tid] = t;
        }
    }
}

#ifdef HELIO_STAMPS
}  // namespace helio
// diagnostic library only: point the stamp buffer of the fused kernel at `stamps_d`
// ([workgroups][KG waves][10] 64-bit words; NULL switches the flush off)
extern "C" int helio_diag_set_stamps(unsigned long long* stamps_d) {
    return hipMemcpyToSymbol(HIP_SYMBOL(helio::g_stamps), &stamps_d, sizeof(stamps_d)) == hipSuccess ? 0 : -1;
}
// … and, for stamping two CONSECUTIVE launches: the size of one launch's stamp block in 64-bit words, and the slot (0 / 1)
// the launches enqueued from now on write (a host-side value carried by each launch's own arguments)
static int g_diag_slot = 0;
extern "C" int helio_diag_set_slot_words(long words) {
    return hipMemcpyToSymbol(HIP_SYMBOL(helio::g_stamp_slot_words), &words, sizeof(words)) == hipSuccess ? 0 : -1;
}
extern "C" void helio_diag_set_slot(int slot) { g_diag_slot = slot & 1; }
namespace helio {
#endif

// A handful of rays per image (the reference's test-time-compute sweeps run ONE heliostat and 500 suns,
// run_experiments.py:31-56): the render is bound by streaming the image out — and, in HelioEnv.step,
// the target image and the distance map in — not by arithmetic, and 32×32 MFMA blocks would each
// re-trace the same ray (the block kernel above ran that shape at 44 % of the HBM rate).  Here a
// workgroup owns a band of FEW_ROWS image rows of one sun: threads 0..N-1 trace the rays once into LDS,
// the band's row factors A_n[i] follow (one exponential per thread), and then a thread keeps ONE 16-byte
// column quad for the whole band — its column factors E_n[j..j+3] live in registers, 4 exponentials per ray
// and thread, once — and walks down the band's rows: per row and quad N broadcast LDS reads, 4·N FMAs and one
// float4 store (lanes ↔ consecutive quads: a wave writes 1 KB of consecutive addresses).  Round 3's form
// evaluated 1 + 4 exponentials per ray and QUAD: at N = 8 that is 40 quarter-rate instructions per 16 bytes,
// compute-bound at a third of the HBM rate; the sums are the same fmaf chains over the same factors (same
// bits).  LOSS = true: HelioEnv.step's forward in the same launch — the band's share of the three image sums
// from the pixels in registers, and band 0 also does the per-ray side work (outputs, the two ray losses,
// the `aux` row).  Partials: [B, bands, 3] / [B, 2], reduced by step_losses_final in fixed order.  Needs
// R % 4 == 0 and 16-byte aligned images; anything else takes the block kernel.
constexpr int FEW_ROWS = 32, FEW_MAX_ROWS = 128, FEW_MAX_RAYS = 8;
// rows per band: a thread's 4·N column-factor exponentials are paid once per band, so narrow images get taller bands
// (R = 128, N = 8, 32 rows: one exponential per pixel quad and ray again — 0.52 of the HBM rate; 128 rows: a quarter
// of that) — as long as the grid keeps ≈2048 workgroups (B = 500, R = 128 stays at 32 rows: 2000 short workgroups hide
// each other's trace and barriers, 500 tall ones would not)
static int few_band_rows(int B, int R) {
    int rows = FEW_MAX_ROWS;
    while (rows > FEW_ROWS && ((long)rows * (R >> 2) > 16 * 256 || (long)B * ((R + rows - 1) / rows) < 2048)) rows >>= 1;
    return rows;
}

// NMAX: 1, 2, 4 or 8 >= N — the column factors are NMAX·4 registers, and with them the kernel fits 8 waves per SIMD
// (its workgroups are short: a band of 32 rows; what hides a workgroup's trace and its two barriers is the other seven)
template <bool LOSS, int NMAX>
__global__ void __launch_bounds__(256, 2)
render_fwd_few(int N, int R, int band_rows, const float* __restrict__ helios, const float* __restrict__ sun,
               const float* __restrict__ action, const float* __restrict__ trig, long trig_b_stride,
               const float* __restrict__ xs, const float* __restrict__ ys, PlaneK P,
               float* __restrict__ actual, float* __restrict__ refl, float* __restrict__ rays,
               float* __restrict__ image, StepLossArgs L) {
    __shared__ float4 sRay[FEW_MAX_RAYS];              // (a, b, k2, c2)
    __shared__ __attribute__((aligned(16))) float sA[FEW_MAX_ROWS][FEW_MAX_RAYS];   // row factors of the band: A_n[i0 + il]
    __shared__ float sLoss[2 * FEW_MAX_RAYS];
    __shared__ float scratch[4];
    const int band = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int i0 = band * band_rows, rows = min(band_rows, R - i0);
    if (tid < N) {
        const int n = tid;
        const long m = (long)b * N + n;
        const vec3 s = ld3(sun + 3l * b);
        const float4 tg = *reinterpret_cast<const float4*>(trig + (long)b * trig_b_stride + 4l * n);
        const vec3 v = ld3(action + 3 * m);
        const Ray q = trace(v, tg.x, tg.y, tg.z, tg.w, ld3(helios + 3l * n), s, P);
        sRay[n] = make_float4(q.a, q.b, q.k2, q.c2);
        if (band == 0) {
            st3(actual + 3 * m, q.act);
            if (refl) st3(refl + 3 * m, q.r);
            if (rays) *reinterpret_cast<float4*>(rays + 4 * m) = make_float4(q.a, q.b, q.k2, q.c2);
            if constexpr (LOSS) {
                const float act[3] = {q.act.x, q.act.y, q.act.z};
                const RayLoss r = ray_loss(L.ideal + 3 * m, act, action + 3 * m, helios + 3l * n, L.g);
                L.align_err[m] = r.ang;
                L.all_bounds[m] = r.out;
                if (L.aux) {     // observation row [sun_b, action_b] (test_environment.py:424)
                    float* a = L.aux + (long)b * (3 + 3l * N);
                    a[3 + 3 * n] = v.x; a[4 + 3 * n] = v.y; a[5 + 3 * n] = v.z;
                    if (n == 0) { a[0] = s.x; a[1] = s.y; a[2] = s.z; }
                }
                sLoss[2 * n] = r.ang;
                sLoss[2 * n + 1] = L.g.exponential_risk ? expf(r.out + 1e-6f) : r.out;
            }
        }
    }
    // (requested while the rays are traced: the row coordinates of this thread's row factors — rows fil, fil + 32, …)
    const int fn = tid & (FEW_MAX_RAYS - 1), fil = tid >> 3;               // 32 rows × FEW_MAX_RAYS = 256 threads
    float xrow[FEW_MAX_ROWS / FEW_ROWS];
#pragma unroll
    for (int k = 0; k < FEW_MAX_ROWS / FEW_ROWS; ++k) xrow[k] = xs[min(i0 + fil + FEW_ROWS * k, R - 1)];
    __syncthreads();
    if constexpr (LOSS) {
        if (band == 0 && tid == 0) {
            float sa = 0.0f, sb = 0.0f;
            for (int n = 0; n < N; ++n) { sa += sLoss[2 * n]; sb += sLoss[2 * n + 1]; }
            L.part_ray[2l * b] = sa;
            L.part_ray[2l * b + 1] = sb;
        }
    }
    if (fn < N) {
        const float4 q = sRay[fn];
#pragma unroll
        for (int k = 0; k < FEW_MAX_ROWS / FEW_ROWS; ++k) {
            if (FEW_ROWS * k < rows) {                 // (uniform)
                const float t = xrow[k] + q.x;
                sA[fil + FEW_ROWS * k][fn] = exp2_fast(-(__builtin_fmaf(t, t, q.w) * q.z));
            }
        }
    }
    const int qpr = R >> 2;                            // 16-byte quads per row
    // thread ↔ (column quad, row phase): rpar rows of the band are walked side by side when a row is shorter than
    // the workgroup (R = 128: 32 quads, 8 rows at a time; consecutive rows are consecutive addresses)
    const int rpar = qpr >= 256 ? 1 : 256 / qpr;
    const int r0 = qpr >= 256 ? 0 : tid / qpr;
    const long base = (long)b * R * R + (long)i0 * R;
    const float sc = LOSS ? L.tx[b] : 1.0f;
    float sq = 0.f, ab = 0.f, ds = 0.f;
    __syncthreads();
    for (int jq = qpr >= 256 ? tid : tid - r0 * qpr; jq < qpr && r0 < rpar; jq += 256) {     // (one pass unless R > 1024)
        const int j = 4 * jq;
        const float4 yj = *reinterpret_cast<const float4*>(ys + j);
        float e[NMAX][4];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
            if (n < N) {                               // (uniform)
                const float4 q = sRay[n];              // broadcast read
                const float u0 = yj.x + q.y, u1 = yj.y + q.y, u2 = yj.z + q.y, u3 = yj.w + q.y;
                e[n][0] = exp2_fast(-((u0 * u0) * q.z)); e[n][1] = exp2_fast(-((u1 * u1) * q.z));
                e[n][2] = exp2_fast(-((u2 * u2) * q.z)); e[n][3] = exp2_fast(-((u3 * u3) * q.z));
            } else {
                e[n][0] = e[n][1] = e[n][2] = e[n][3] = 0.0f;
            }
        }
        for (int il = r0; il < rows; il += rpar) {
            const long p = base + (long)il * R + j;
            float4 tg4 = make_float4(0.f, 0.f, 0.f, 0.f), dm4 = tg4;
            if constexpr (LOSS) {                      // requested first: they are what the kernel waits for
                tg4 = *reinterpret_cast<const float4*>(L.target + p);
                dm4 = *reinterpret_cast<const float4*>(L.dmaps + p);
            }
            const float4 a03 = *reinterpret_cast<const float4*>(&sA[il][0]);      // broadcast reads
            float4 a47 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (NMAX > 4) a47 = *reinterpret_cast<const float4*>(&sA[il][4]);
            const float av[FEW_MAX_RAYS] = {a03.x, a03.y, a03.z, a03.w, a47.x, a47.y, a47.z, a47.w};
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                if (n < N) {                           // (uniform; rays past N hold stale factors: never read)
                    acc.x = __builtin_fmaf(av[n], e[n][0], acc.x);
                    acc.y = __builtin_fmaf(av[n], e[n][1], acc.y);
                    acc.z = __builtin_fmaf(av[n], e[n][2], acc.z);
                    acc.w = __builtin_fmaf(av[n], e[n][3], acc.w);
                }
            }
            *reinterpret_cast<float4*>(image + p) = acc;
            if constexpr (LOSS) {
                const float pv[4] = {acc.x, acc.y, acc.z, acc.w}, tv[4] = {tg4.x, tg4.y, tg4.z, tg4.w};
                const float dv[4] = {dm4.x, dm4.y, dm4.z, dm4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = pv[k] / sc - tv[k] / sc;       // as the reference divides (:438-441)
                    const float ad = fabsf(d);
                    sq = __builtin_fmaf(d, d, sq);
                    ab += ad;
                    ds = __builtin_fmaf(ad, dv[k], ds);
                }
            }
        }
    }
    if constexpr (LOSS) {
        sq = block_sum(sq, scratch);
        ab = block_sum(ab, scratch);
        ds = block_sum(ds, scratch);
        if (tid == 0) {
            float* o = L.part_img + 3l * ((long)b * gridDim.x + band);
            o[0] = sq; o[1] = ab; o[2] = ds;
        }
    }
}

template <bool LOSS>
static void launch_few(int B, int N, int R, const float* helios, const float* sun, const float* action, const float* trig,
                       long trig_b_stride, const float* xs, const float* ys, const PlaneK& P, float* actual, float* refl,
                       float* rays, float* image, const StepLossArgs& L, hipStream_t st) {
    const int band_rows = few_band_rows(B, R);
    const dim3 grid((R + band_rows - 1) / band_rows, B), block(256);
#define HELIO_FEW_FWD(NM) hipLaunchKernelGGL((render_fwd_few<LOSS, NM>), grid, block, 0, st, N, R, band_rows, helios, sun, action, trig, trig_b_stride, xs, ys, P, actual, refl, rays, image, L)
    if (N <= 1) HELIO_FEW_FWD(1);
    else if (N <= 2) HELIO_FEW_FWD(2);
    else if (N <= 4) HELIO_FEW_FWD(4);
    else HELIO_FEW_FWD(8);
#undef HELIO_FEW_FWD
}

// the streaming kernel's preconditions
static bool few_ok(int N, int R, const float* ys, const float* image, const StepLossArgs* L) {
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return N <= FEW_MAX_RAYS && (R & 3) == 0 && a16(ys) && a16(image) && (!L || (a16(L->target) && a16(L->dmaps)));
}
// and where it wins: always for one or two rays; for up to 8 once the images are large enough to be
// bound by HBM rather than by launch latency (tools/bench_fused.py: N = 8, B = 64, R = 64 takes 4.4 µs in
// the block kernel and 5.3 µs here — 40 exponentials per pixel quad)
// (round 4, tools/rule_regret.py as graph replays: below ≈0.5 M pixels even one or two rays are 0.3–0.7 µs quicker in the
// block kernel — B = 4, N = 2, R = 256: 3.6 against 4.3 µs — whose single barrier is the shorter latency chain)
// Final form, from the second table (profiles/r04_c_rule_regret.txt): one or two rays — from 64 bands of 32 rows (B = 25,
// R = 100: 3.9 against 4.3 µs; B = 4, R = 256, 32 bands: 4.1 / 3.6); up to four — from 2 M pixels; up to eight — from 8 M
// (B = 500, N = 8, R = 128, 8.2 M: 13.0 against 11.8 µs for geometry + the 64² register kernel; R = 256, 33 M: level).
static bool few_wins(int B, int N, int R) {
    if (N <= 2) return (long)B * ((R + FEW_ROWS - 1) / FEW_ROWS) >= 64;
    return (long)B * R * R >= (N <= 4 ? (1l << 21) : (1l << 23));
}

// true when launch_render_fwd() would take the single-launch path
bool render_is_fused(int B, int N, int R) {
    // one launch instead of two wherever the launch boundary (≈1.5–2 µs plus a second wait for kernel
    // arguments) is a visible share of the render: up to 512 128²-tiles' worth of pixels and a heliostat
    // sum that one workgroup traces in one go (N <= 256: four waves of 64 rays).  Every 32×32 block
    // traces the rays of its sun itself, so longer sums go to the geometry + splat pair
    // a handful of rays per image: one streaming launch at any size (render_fwd_few, or the block kernel
    // when its alignment preconditions do not hold)
    // … and only while the re-tracing stays small: every 32×32 block traces all N rays of its sun, B·N·⌈R/32⌉² traces in
    // all against the B·N of the geometry kernel.  tools/rule_regret.py (profiles/r04_c_rule_regret_before.txt): from
    // ≈300 k traces the two launches win — B = 500, N = 96, R = 128: 35.8 against 24.3 µs; B = 256, N = 200, R = 128:
    // 36.1 / 27.3; B = 60, N = 200, R = 256: 34.4 / 27.8; at 256 k (B = 500, N = 32, R = 128) still by 11 %: 16.8 / 15.2; at
    // 205 k (B = 256, N = 50, R = 128) the single launch is level or ahead
    const long t128 = (long)B * ((R + 127) / 128) * ((R + 127) / 128);
    const long nb = (R + 31) / 32;
    return N <= FEW_MAX_RAYS || (N <= 256 && t128 < 512 && (long)B * N * nb * nb <= 225000);
}

// … for the render ALONE (helio_render_fwd; the env step keeps the rule above: its fused forms also spare the loss block a
// pass over the image).  A handful of rays no longer means one launch at any size: where the streaming kernel does not
// win (few_wins) the block kernel must earn its place like everybody — and with more than 4096 blocks it does not, however
// few rays each traces: B = 500, N = 8, R = 100: 12.6 against 10.6 µs for geometry + the 128² register kernel; N = 16,
// R = 128: 14.0 / 13.2 (third table, profiles/r04_c_rule_regret.txt)
bool render_is_fused_plain(int B, int N, int R) {
    if (N <= FEW_MAX_RAYS && few_wins(B, N, R)) return true;
    const long t128 = (long)B * ((R + 127) / 128) * ((R + 127) / 128);
    const long nb = (R + 31) / 32;
    return N <= 256 && t128 < 512 && (long)B * N * nb * nb <= 225000 && (long)B * nb * nb <= 4096;
}

// waves per 32×32 block of the fused kernel (the heliostats are split between them; 64·KG >= N): as
// many as keep the chip's 1024 SIMDs at about one wave each and leave every wave a few k-pairs.
// HELIO_FUSED_KG (1, 2 or 4) forces one form where N allows it — tuning runs only
int fused_kg(int B, int N, int R) {
    static const int forced = [] { const char* e = getenv("HELIO_FUSED_KG"); return e ? atoi(e) : 0; }();
    const int need = N > 128 ? 4 : (N > 64 ? 2 : 1);
    if ((forced == 1 || forced == 2 || forced == 4) && forced >= need) return forced;
    const long nb = (R + 31) / 32, blocks = (long)B * nb * nb;
    const int pairs = (N + 1) / 2;
    int kg = 1;
    if (pairs >= 8 && blocks * 4 <= 2048) kg = 4;
    else if (pairs >= 4 && blocks * 2 <= 2048) kg = 2;
    return kg > need ? kg : need;
}

#ifdef HELIO_STAMPS
}  // namespace helio
extern "C" int helio_diag_fused_kg(int B, int N, int R) { return helio::fused_kg(B, N, R); }
namespace helio {
#endif

#ifdef HELIO_STAMPS
#define HELIO_DIAG_SLOT_BIT (::g_diag_slot << 17)
#else
#define HELIO_DIAG_SLOT_BIT 0
#endif

template <int KG, bool LOSS>
static void launch_fused(int B, int N, int R, const float* helios, const float* sun, const float* action,
                         const float* trig, long trig_b_stride, const helio_plane* plane, const float* xs,
                         const float* ys, float* actual, float* refl, float* rays, float* image,
                         const StepLossArgs& L, hipStream_t st) {
    const int nb = (R + 31) / 32;
    FusedLate late;
    late.P = to_k(plane);
    late.actual = actual; late.refl = refl; late.rays = rays; late.image = image;
    hipLaunchKernelGGL((render_fwd_fused_small<KG, LOSS>), dim3(nb, nb + (LOSS ? 1 : 0), B), dim3(64 * KG), 0, st, N,
                       R | (trig_b_stride != 0 ? 1 << 16 : 0) | HELIO_DIAG_SLOT_BIT, helios, sun, action, trig, xs, ys, late, L);
}

// form: 0 = by problem size; 1, 2, 4 = the block kernel with that many waves per block (64·KG >= N);
// 8 = the few-ray streaming kernel (N <= 8, R % 4 == 0, 16-byte aligned images) — forced forms are for
// the parity tests and tuning runs.  → false when the forced form does not exist for this problem
static int resolve_fused_form(int form, int B, int N, int R, bool few_possible) {
    if (form == 0) return few_possible && few_wins(B, N, R) ? 8 : fused_kg(B, N, R);
    if (form == 8) return few_possible ? 8 : -1;
    if ((form == 1 || form == 2 || form == 4) && N <= 64 * form) return form;
    return -1;
}

bool launch_render_fused(int B, int N, int R, const float* helios, const float* sun, const float* action,
                         const float* trig, long trig_b_stride, const helio_plane* plane, const float* xs,
                         const float* ys, float* actual, float* refl, float* rays, float* image, int form,
                         hipStream_t st) {
    const StepLossArgs none{};
    switch (resolve_fused_form(form, B, N, R, few_ok(N, R, ys, image, nullptr))) {
    case 8:
        launch_few<false>(B, N, R, helios, sun, action, trig, trig_b_stride, xs, ys, to_k(plane), actual, refl, rays, image, none, st);
        return true;
    case 4: launch_fused<4, false>(B, N, R, helios, sun, action, trig, trig_b_stride, plane, xs, ys, actual, refl, rays, image, none, st); return true;
    case 2: launch_fused<2, false>(B, N, R, helios, sun, action, trig, trig_b_stride, plane, xs, ys, actual, refl, rays, image, none, st); return true;
    case 1: launch_fused<1, false>(B, N, R, helios, sun, action, trig, trig_b_stride, plane, xs, ys, actual, refl, rays, image, none, st); return true;
    default: return false;
    }
}

void launch_step_losses_final(int, int, int, int, int, float, const float*, const float*, float*, float*, float*,
                              int*, int, hipStream_t);

// workspace floats of the fused env step: [B, blocks, 3] image partials + [B, 2] ray partials
long env_step_fused_workspace(int B, int R) {
    const long t = (R + 31) / 32;
    return 3l * B * t * t + 2l * B;
}

// HelioEnv.step forward for the problems render_is_fused() selects: render + loss partials in one
// launch, then the finishing workgroup of step_losses.hip — 2 launches for the whole step
bool launch_env_step_fused(int B, int N, int R, const float* helios, const float* sun, const float* action,
                           const float* trig, long trig_b_stride, const helio_plane* plane, const float* xs,
                           const float* ys, float* actual, float* refl, float* rays, float* image,
                           const float* target, const float* tx, const float* dmaps, const float* ideal,
                           const float* tp, const float* tn, float W, float H, int exponential_risk,
                           float mask_ratio, float* workspace, float* out, float* mae, float* keep,
                           float* align_err, float* all_bounds, float* aux, int* notify, int ticket, int form,
                           hipStream_t st) {
    const int t = (R + 31) / 32;
    StepLossArgs L;
    L.target = target; L.tx = tx; L.dmaps = dmaps; L.ideal = ideal;
    L.part_img = workspace; L.part_ray = workspace + 3l * B * t * t;
    L.align_err = align_err; L.all_bounds = all_bounds; L.aux = aux;
    L.g = make_geom(tp, tn, W, H, exponential_risk);
    switch (resolve_fused_form(form, B, N, R, few_ok(N, R, ys, image, &L))) {
    case 8: {
        const int band_rows = few_band_rows(B, R), bands = (R + band_rows - 1) / band_rows;      // (launch_few's grid)
        L.part_ray = workspace + 3l * B * bands;
        launch_few<true>(B, N, R, helios, sun, action, trig, trig_b_stride, xs, ys, to_k(plane), actual, refl, rays, image, L, st);
        launch_step_losses_final(B, N, R, bands, B, mask_ratio, L.part_img, L.part_ray, out, mae, keep, notify, ticket, st);
        return true;
    }
    case 4: launch_fused<4, true>(B, N, R, helios, sun, action, trig, trig_b_stride, plane, xs, ys, actual, refl, rays, image, L, st); break;
    case 2: launch_fused<2, true>(B, N, R, helios, sun, action, trig, trig_b_stride, plane, xs, ys, actual, refl, rays, image, L, st); break;
    case 1: launch_fused<1, true>(B, N, R, helios, sun, action, trig, trig_b_stride, plane, xs, ys, actual, refl, rays, image, L, st); break;
    default: return false;
    }
    launch_step_losses_final(B, N, R, t * t, B, mask_ratio, L.part_img, L.part_ray, out, mae, keep, notify, ticket, st);
    return true;
}

// (A double-buffered form — 32-ray chunks, two LDS buffers, one barrier per chunk, producers
// overlapping other waves' MFMAs — was measured at 114 TFLOP/s against 133.5 for the two-phase
// kernel above: halving the chunk keeps the barrier rate per MFMA and doubles the per-chunk
// fixed costs.  Not kept.)

template <int MBI, int MBJ, int WI, int WJ, int NC, bool TWO_LEVEL>
static void launch_regs(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                        float* image, hipStream_t st, CullFwd cull = CullFwd{nullptr, nullptr, nullptr}) {
    constexpr int TI = 32 * MBI * WI, TJ = 32 * MBJ * WJ;
    const int ti = (R + TI - 1) / TI, tj = (R + TJ - 1) / TJ;
    hipLaunchKernelGGL((splat_fwd_mfma_regs<MBI, MBJ, WI, WJ, NC, TWO_LEVEL>), dim3(ti * tj, B), dim3(256), 0, st,
                       B, N, R, rays, xs, ys, image, cull.counts, cull.order, cull.lists);
}

template <int W, bool TWO_LEVEL>
static void launch_tile(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                        float* image, hipStream_t st, CullFwd cull = CullFwd{nullptr, nullptr, nullptr}, int S = 1, int P = 0,
                        float* partials = nullptr) {
    constexpr int T = 64 * W;
    const int t = (R + T - 1) / T;
    const size_t lds = (2 * 64 * (T + (W == 4 ? 2 : 1)) + 2 * T) * sizeof(float);
    static bool configured = false;   // raising the dynamic-LDS cap is idempotent; a race is harmless
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(splat_fwd_mfma_tile<W, TWO_LEVEL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    hipLaunchKernelGGL((splat_fwd_mfma_tile<W, TWO_LEVEL>), dim3(t * t * S, B), dim3(64 * W * W), lds, st,
                       B, N, R, rays, xs, ys, image, cull.counts, cull.order, cull.lists, S, P, partials);
}


// ----------------------------------------------------------------------------------------------
// Split-bf16 variant (opt-in, variant 7): the same rank-N outer-product sum on the bf16 matrix
// pipe, which on gfx950 runs 16× the f32 MFMA rate.  Every f32 factor is split EXACTLY into three
// bf16 pieces by truncation (hi = top 8 significant bits, mid = the next 8, lo = the last 8: 24 bits,
// so A = Ah + Am + Al with no rounding), and a product A·E is issued as the six partial products
//   Ah·Eh + Ah·Em + Am·Eh + Ah·El + Al·Eh + Am·Em
// on v_mfma_f32_32x32x16_bf16 (bf16×bf16 is exact in f32; accumulation stays f32).  What is dropped,
// Am·El + Al·Em + Al·El, is below 2^-23 of the product: an error of the size of one f32 rounding
// (always towards zero), where the exact-f32 kernels above make none in the product and one in the
// add.  Six MFMAs of 32 cycles do the work of eight f32 MFMAs of 64: 2.7× fewer matrix-pipe cycles,
// and — unlike the f32 MFMA — the bf16 MFMA holds the SIMD's vector issue for only 8 of its 32
// cycles, so the factor evaluation of the NEXT chunk hides under the MFMAs of the current one.
//
// Measured at config 4 (N=2000, B=512, R=512), against 3.96 ms for splat_fwd_mfma_tile<4>:
//   variant 7 (two-level sums, below): 2.24 ms; against fp64 (tools/accuracy_splat.py) worst per-pixel
//     relative error 8.6e-7, rms 1.6e-7 — tighter than tile<4>'s one-level f32 chain (1.2e-6 / 2.1e-7);
//   variant 8 (one level): 1.95 ms; worst pixel 2.5e-6, mean -6e-7 (the pipe's truncation, see below).
// The chip runs these kernels at 2.06 GHz (2.34 under the f32 MFMA); in variant 8 the bf16 pipe is
// busy 72–75 % of those cycles (PMC), the rest being the barrier and the first operand fetch of every
// 16-ray trip.  Both pass every parity fixture at 1e-5; neither is the literal f32 fmaf chain of the
// other kernels, hence opt-in.
//
// Workgroup = 8 waves = one 256×256 tile (2×4 waves of 128×64 pixels, 8 accumulator blocks each);
// chunk = 16 rays = one k-step; double-buffered LDS tables in MFMA operand order
// T[piece][k-half][pixel][8 × bf16] (consecutive lanes ↔ consecutive 16-byte slots: conflict-free
// ds_write_b128 / ds_read_b128); thread p of the workgroup owns pixel p of the tile's 256 rows + 256
// columns and produces its 16 factors of the next chunk; one barrier per chunk.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack_hi16(float lo_elem, float hi_elem) {
    // bf16(lo_elem) | bf16(hi_elem) << 16, both by truncation (v_perm_b32)
    return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}

template <bool TWO_LEVEL>
__global__ void __launch_bounds__(512)
splat_fwd_mfma_bf16x3(int B, int N, int R, const float* __restrict__ rays, const float* __restrict__ xs,
                      const float* __restrict__ ys, float* __restrict__ image) {
    constexpr int KC = 16, T = 256;
    constexpr int PIECE = 2 * T * 16;                 // bytes of one piece table: [half][pixel][16 B]
    constexpr int TABLE = 3 * PIECE;                  // A or E, three pieces
    constexpr int BUF = 2 * TABLE;                    // A then E
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2·BUF + ray tables
    float4* sRay = reinterpret_cast<float4*>(lds + 2 * BUF);              // [2 buffers][KC][row/col]

    const int tiles = (R + T - 1) / T;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int ti0 = (blockIdx.x / tiles) * T, tj0 = (blockIdx.x % tiles) * T;
    const int wi = (wave >> 2) * 128, wj = (wave & 3) * 64;

    // producer role: pixel p of the tile (rows 0..255, then columns 0..255)
    const bool isrow = tid < T;
    const int pp = isrow ? tid : tid - T;
    const float coord = isrow ? xs[min(ti0 + pp, R - 1)] : ys[min(tj0 + pp, R - 1)];
    const int ptab = (isrow ? 0 : TABLE) + pp * 16;
    const int rsel = isrow ? 0 : 1;

    // ray parameters of a chunk: fetched from global memory two trips before they are staged (the load
    // latency never sits between two barriers), pre-scaled and written to LDS by 16 lanes of wave 0
    auto fetch_rays = [&](int chunk) -> float4 {
        const int n = min(chunk * KC + (tid & (KC - 1)), N - 1);
        return reinterpret_cast<const float4*>(rays)[(long)b * N + n];
    };
    auto stage_rays = [&](float4 q, int chunk, int buf) {
        // lanes 0..15 of wave 0 (measured: 1.98 ms; the same on a vector-phase-first wave: 2.09 ms)
        if (tid < KC) {
            const int n = chunk * KC + tid;
            float4 rowp = make_float4(0.f, 0.f, 1e30f, 0.f), colp = make_float4(0.f, 0.f, 0.f, 0.f);   // padding: A = 0
            if (n < N) {
                const float sk = __builtin_sqrtf(q.z);
                rowp = make_float4(q.x * sk, sk, q.w * q.z, 0.f);
                colp = make_float4(q.y * sk, sk, 0.f, 0.f);
            }
            sRay[(buf * KC + tid) * 2] = rowp;
            sRay[(buf * KC + tid) * 2 + 1] = colp;
        }
    };
    auto produce = [&](int buf) {                     // this thread's 16 factors of the chunk staged in sRay[buf]
        unsigned char* base = lds + buf * BUF + ptab;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float hi[8], mid[8], lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 r = sRay[(buf * KC + 8 * h + j) * 2 + rsel];     // (shift·sk, sk, cc): a broadcast read
                const float q = __builtin_fmaf(coord, r.y, r.x);
                const float f = exp2_fast(-__builtin_fmaf(q, q, r.z));
                hi[j] = __uint_as_float(__float_as_uint(f) & 0xFFFF0000u);
                const float r1 = f - hi[j];
                mid[j] = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
                lo[j] = r1 - mid[j];
            }
            uint4 vh = make_uint4(pack_hi16(hi[0], hi[1]), pack_hi16(hi[2], hi[3]), pack_hi16(hi[4], hi[5]), pack_hi16(hi[6], hi[7]));
            uint4 vm = make_uint4(pack_hi16(mid[0], mid[1]), pack_hi16(mid[2], mid[3]), pack_hi16(mid[4], mid[5]), pack_hi16(mid[6], mid[7]));
            uint4 vl = make_uint4(pack_hi16(lo[0], lo[1]), pack_hi16(lo[2], lo[3]), pack_hi16(lo[4], lo[5]), pack_hi16(lo[6], lo[7]));
            *reinterpret_cast<uint4*>(base + 0 * PIECE + h * T * 16) = vh;
            *reinterpret_cast<uint4*>(base + 1 * PIECE + h * T * 16) = vm;
            *reinterpret_cast<uint4*>(base + 2 * PIECE + h * T * 16) = vl;
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rb][cb][e] = 0.0f;

    const int chunks = (N + KC - 1) / KC;
    stage_rays(fetch_rays(0), 0, 0);
    float4 qnext = fetch_rays(1);
    __syncthreads();
    produce(0);
    stage_rays(qnext, 1, 1);                           // (chunks past the end stage padding rays: A = 0)
    qnext = fetch_rays(2);
    // consumer operand addresses: piece P, half lh, pixel (wave offset + 32·block + lr)
    const int offA = (lh * T + wi + lr) * 16, offE = TABLE + (lh * T + wj + lr) * 16;
    for (int c = 0; c < chunks; ++c) {
        __syncthreads();                               // tables[c&1] complete; rays of chunk c+1 staged
        const int buf = c & 1;
        stage_rays(qnext, c + 2, buf);                 // sRay[buf] was read by produce() of the previous trip
        qnext = fetch_rays(c + 3);
        // The factors of chunk c+1 (VALU, into the other buffer; the last trip produces an unused
        // padding chunk) and the 48 MFMAs of chunk c are independent.  The two waves that share a SIMD
        // (w and w+4) run them in OPPOSITE order: while one wave's MFMAs occupy the matrix pipe the
        // other wave's factor evaluation issues on the vector ALU (a bf16 MFMA holds the vector issue
        // for 8 of its 32 cycles), then they swap — the pipe never waits for a producer phase.
        auto consume = [&]() {
            const unsigned char* tb = lds + buf * BUF;
            if constexpr (!TWO_LEVEL) {
                // all 18 operand fragments first (72 VGPRs), in the order the MFMAs need them
                bf16x8 ep[3][2], ap[4][3];
#pragma unroll
                for (int P = 0; P < 3; ++P)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) ep[P][cb] = *reinterpret_cast<const bf16x8*>(tb + offE + P * PIECE + cb * 512);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int P = 0; P < 3; ++P) ap[rb][P] = *reinterpret_cast<const bf16x8*>(tb + offA + P * PIECE + rb * 512);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        f32x16 v = acc[rb][cb];
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[rb][1], ep[1][cb], v, 0, 0, 0);      // small terms first
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[rb][2], ep[0][cb], v, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[rb][0], ep[2][cb], v, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[rb][1], ep[0][cb], v, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[rb][0], ep[1][cb], v, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[rb][0], ep[0][cb], v, 0, 0, 0);
                        acc[rb][cb] = v;
                    }
                }
            } else {
                // Two-level sums.  The bf16 pipe aligns the 16 products of an instruction against the
                // accumulator's exponent and truncates: against the running total of N = 2000 rays that is
                // a bias of -6e-7 (worst pixel 2.5e-6); against a ZERO accumulator holding one 16-ray
                // partial sum it is ~1/125 of that.  So the six partial products of a block go into a
                // fresh accumulator and the partial sum is added to the running total on the vector ALU,
                // round-to-nearest — under the MFMAs of the next row block.  Measured against fp64: worst
                // pixel 8.6e-7, rms 1.6e-7 — tighter than the one-level f32 chain of splat_fwd_mfma_tile<4>
                // (1.2e-6 / 2.1e-7); the adds cost vector issue slots: 2.24 ms instead of 1.95.
                bf16x8 ep[3][2];
#pragma unroll
                for (int P = 0; P < 3; ++P)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) ep[P][cb] = *reinterpret_cast<const bf16x8*>(tb + offE + P * PIECE + cb * 512);
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                f32x16 part[2][2];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    bf16x8 ap[3];
#pragma unroll
                    for (int P = 0; P < 3; ++P) ap[P] = *reinterpret_cast<const bf16x8*>(tb + offA + P * PIECE + rb * 512);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        f32x16 v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], ep[1][cb], zero, 0, 0, 0);      // small terms first
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], ep[0][cb], v, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], ep[2][cb], v, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], ep[0][cb], v, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], ep[1][cb], v, 0, 0, 0);
                        part[rb & 1][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], ep[0][cb], v, 0, 0, 0);
                    }
                    if (rb > 0) {
                        acc[rb - 1][0] += part[(rb - 1) & 1][0];
                        acc[rb - 1][1] += part[(rb - 1) & 1][1];
                        // the adds stay HERE: left alone they are sunk into the next trip and eight partial
                        // sums stay live (256 VGPRs + spills)
                        asm volatile("" : "+v"(acc[rb - 1][0]), "+v"(acc[rb - 1][1]));
                        // ... and go between this row block's MFMAs, two per MFMA slot, not after them
#pragma unroll
                        for (int i = 0; i < 12; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[3][0] += part[1][0];
                acc[3][1] += part[1][1];
                asm volatile("" : "+v"(acc[3][0]), "+v"(acc[3][1]));
            }
        };
        if (wave & 4) {          // measured: 1.98 ms; skew by wave&1 / wave&2 / none: 2.29 / 2.24 / 2.26 ms
            produce(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            consume();
        } else {
            consume();
            __builtin_amdgcn_sched_barrier(0);
            produce(buf ^ 1);
        }
    }
    float* img = image + (long)b * R * R;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            store_block(img, R, ti0 + wi + 32 * rb, tj0 + wj + 32 * cb, lr, lh, acc[rb][cb]);
}

template <bool TWO_LEVEL>
static void launch_bf16x3(int B, int N, int R, const float* rays, const float* xs, const float* ys, float* image,
                          hipStream_t st) {
    const int t = (R + 255) / 256;
    const size_t lds = 2 * (2 * 3 * 2 * 256 * 16) + 2 * 16 * 2 * sizeof(float4);
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(splat_fwd_mfma_bf16x3<TWO_LEVEL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    hipLaunchKernelGGL(splat_fwd_mfma_bf16x3<TWO_LEVEL>, dim3(t * t, B), dim3(512), lds, st, B, N, R, rays, xs, ys, image);
}

// variant: 0/2 = MFMA, kernel chosen by problem size; 1 = VALU; 3/4/5/6 force one MFMA kernel
// (regs 128², tile 128², tile 256², regs 64²) — used by the tests and tools/bench_splat.py.
// (tile 128² is never the fastest in the sweep; it stays as a forced variant for A/B runs.)
// the kernel variant 0 stands for at this size (a forced variant gives the same bits for ANY number of
// images: every kernel sums an image's heliostats in an order that depends on N and R only)
static int splat_fwd_choice(int B, int N, int R) {
    const long t128 = (long)B * ((R + 127) / 128) * ((R + 127) / 128);
    const long t256 = (long)B * ((R + 255) / 256) * ((R + 255) / 256);
    // measured on MI355X over B ∈ {4..256}, N ∈ {50, 500, 5000}, R ∈ {64..512}
    // (tools/sweep_variants.py, tools/sweep_render.py): the 256² LDS-table kernel wins once its tiles fill
    // the 256 CUs and the heliostat sum is long enough to amortise its 64-ray chunks; the 128²
    // register-operand kernel once ITS tiles fill the chip; below that, 32² blocks with the heliostat sum
    // split over the waves of a workgroup where it is long, else 64² tiles.
    // (round 4) … from TWO rounds of the chip: 512 tiles.  Below that a partly filled round idles (B = 384, N = 5000, R = 256: 2416 µs
    // against 1923 with the sum split in two) and — at the reference's default error scale, where the lists bite — the
    // lists do not balance: every workgroup starts at once and the launch lasts as long as its longest list (B = 256,
    // N = 5000, R = 256 at err 90 mrad / σs 0.01: 1050 µs, live fraction 0.60, against 822 for the 128² tiles, 0.50).
    if (N >= 200 && R > 128 && t256 >= 512) return 5;
    // between 24 and 512 such tiles — tens to hundreds of images of a large field — the 256² kernel with the heliostat sum
    // split across workgroups (variants 14..17, below): as many parts as give 512 workgroups, while a part keeps at
    // least 7 of the kernel's 64-ray chunks (16 where the unsplit kernel is the alternative).  tools/sweep_split.py
    // (profiles/r03_c_sweep_split.txt): B = 32, N = 5000, R = 256: 256 → 215 µs (k-split blocks before); B = 64: 464 → 371 µs
    // (128² register tiles before); B = 16, N = 5000, R = 512: 458 → 357 µs.  Round 4, both regimes side by side
    // (tools/rule_regret.py sizes=… with and without err90, profiles/r04_x_split_both_regimes.txt): parts × tiles ≥ 512
    // instead of ≥ 192 costs 2–7 % with every ray live and gains 12–22 % at err 90 (B = 128, N = 5000, R = 256: 565 → 463 µs;
    // B = 32, N = 5000, R = 512: 470 → 386); where parts that short are not to be had and the 128² tiles fill the chip,
    // those (B = 32, N = 1000, R = 512 at err 90: 126 → 101 µs).  HELIO_SPLIT=0 switches the choice off.
    static const bool no_split = [] { const char* e = getenv("HELIO_SPLIT"); return e && e[0] == '0'; }();
    if (!no_split && R > 128 && t256 >= 24 && t256 < 512) {
        int want = 2;
        while (t256 * want < 512 && want < 16) want *= 2;
        const int min_part = t256 >= 192 ? 1000 : 448;
        // (one step fewer parts where the parts would be too short — never fewer than round 3's rule gave: 2 / 4 / 8 parts
        // from 96 / 48 / 24 tiles; B = 32, N = 1000, R = 256 must stay with the k-split blocks: 53 µs against 144 in two parts)
        const int least = t256 >= 96 ? 2 : t256 >= 48 ? 4 : 8;
        int S = want;
        if (N / S < min_part && S / 2 >= least) S /= 2;
        if (N / S >= min_part && (S == want || t128 < 192 || R <= 64))
            return S == 2 ? 14 : S == 4 ? 15 : S == 8 ? 16 : 17;
    }
    if (N >= 200 && R > 128 && t256 >= 192) return 5;
    if (t128 >= 192 && R > 64) return 3;
    return ksplit_parts(B, N, R) ? 9 : 6;
}

// What helio_render_fwd's variant 0 resolves to for (B, N, R): 10..13 (a form of the single-launch kernel)
// or 3, 5, 6, 9, 14..16 (geometry + that splat kernel).  A caller that renders a batch in pieces — one shard per
// GPU — passes the choice of the WHOLE batch with every piece and gets the rows of the unsharded render
// bit for bit.  (The few-ray form assumes 16-byte aligned images, as torch's allocations are.)
int render_fwd_choice(int B, int N, int R) {
    if (render_is_fused_plain(B, N, R)) {
        const int f = resolve_fused_form(0, B, N, R, N <= FEW_MAX_RAYS && (R & 3) == 0);
        return f == 8 ? 13 : f == 4 ? 12 : f == 2 ? 11 : 10;
    }
    return splat_fwd_choice(B, N, R);
}

// Variants 14..17: the 256² LDS-table kernel with the heliostat sum SPLIT into S = 2, 4, 8, 16 parts of P
// consecutive rays (P a multiple of the kernel's 64-ray chunk): few images of many heliostats do not give the
// chip 256 tiles, but tiles × parts do.  A part is summed from zero by a workgroup of its own into a partial
// image in the caller's scratch, and splat_reduce_parts adds the S partial images in part order — the bits are
// a function of N, R and S only, so a shard of a batch forced to the whole batch's variant reproduces its rows.
// These variants NEED the scratch (HELIO_E_SCRATCH without it): the partial images live there.
static int split_parts(int variant) { return variant >= 14 && variant <= 17 ? 2 << (variant - 14) : 1; }
static int split_part_rays(int N, int S) { return (((N + S - 1) / S) + 63) & ~63; }
static long split_partial_bytes(int B, int R, int S) { return S > 1 ? cull_pad256(4l * B * S * R * R) : 0; }

// Skipping exactly-zero rays (cull.h): which lists a variant's kernel takes.  The one-level LDS-table /
// register-operand kernels: one list per (image, tile[, part of a split sum]) — removing terms that leave the
// accumulator unchanged from ONE fmaf chain leaves its result.  The k-split block kernel: one list per (image, part)
// for the whole image (its 16 parts are chains from zero over fixed ray ranges, culled inside each range; every
// 32×32 block of the image walks the same lists) — from N = 1024 and 2^31 (ray, pixel) pairs: it serves few images,
// where the compaction launch must be paid by one short kernel.  (The 64² two-level kernel rounds at 128-ray chunk boundaries — one list per
// chunk would be more lists than rays saved —, the split-bf16 kernels align products against the accumulator inside
// the pipe, the VALU kernels are reference forms: they stay dense.)  From N = 192: below that the compaction launch
// costs what it saves.
struct CullPlan { int te, S, P; bool order; };       // te = 0: this call runs dense
constexpr int CULL_WHOLE_IMAGE = 16384;              // a tile edge no image exceeds (R <= 16384)
static CullPlan cull_fwd_plan(int variant, int B, int N, int R) {
    if (variant == 0 || variant == 2) variant = splat_fwd_choice(B, N, R);
    // lists from 192 rays — and from 64 where the call is long enough to carry the two launches in front (round 4,
    // tools/bench_fwd_tiles.py at both error scales, profiles/r04_w_fwd_lists_small_fields.txt: thousands of suns over a field of
    // 96–128 heliostats at err 90 mrad: B = 1024, N = 128, R = 256: 172 → 116 µs; B = 500, N = 96, R = 256: 73 → 56; with every
    // ray live the lists cost 8–12 % there; below 64 rays what they save and what they cost are level, 10–16 % either way)
    if (!cull_enabled() || N < 64 || (N < 192 && (long)B * N * R * R < (1l << 31))) return {0, 1, 0, false};
    const int S = split_parts(variant);
    // the work order (cull.h) matters once there are more lists than workgroups the chip starts at once; below
    // that its launch is only latency
    const long t256 = (long)B * ((R + 255) / 256) * ((R + 255) / 256) * S, t128 = (long)B * ((R + 127) / 128) * ((R + 127) / 128);
    if (variant == 5 || S > 1) return {256, S, split_part_rays(N, S), t256 > 256};
    if (variant == 3 || variant == 4) return {128, 1, 0, t128 > 256};
    // (tools/sweep_split.py at err 90 / sigma 0.01, k-split with and without lists: B=4, N=5000, R=512: 130 → 86 µs; B=2:
    // 69 → 46; B=8, N=2000, R=512: 102 → 70; nothing at N = 1000; with every ray live the extra launch costs ≈5 µs)
    if (variant == 9 && N >= 1024 && (long)B * N * R * R >= (1l << 31))
        return {CULL_WHOLE_IMAGE, KSPLIT_PARTS, (((N + KSPLIT_PARTS - 1) / KSPLIT_PARTS) + 3) & ~3, false};
    return {0, 1, 0, false};
}

// bytes a call can use (partial images of a split sum + the lists); and the part of it the call cannot do without
long splat_fwd_scratch_bytes(int B, int N, int R, int variant) {
    if (variant == 0 || variant == 2) variant = splat_fwd_choice(B, N, R);
    const CullPlan c = cull_fwd_plan(variant, B, N, R);
    return split_partial_bytes(B, R, split_parts(variant)) + (c.te ? cull_fwd_bytes(B, N, R, c.te, c.S, c.P) : 0);
}
long splat_fwd_scratch_required(int B, int N, int R, int variant) {
    if (variant == 0 || variant == 2) variant = splat_fwd_choice(B, N, R);
    return split_partial_bytes(B, R, split_parts(variant));
}

int launch_splat_fwd(int B, int N, int R, const float* rays, const float* xs, const float* ys,
                     float* image, int variant, void* scratch, long scratch_bytes, hipStream_t st) {
    const long t128 = (long)B * ((R + 127) / 128) * ((R + 127) / 128);
    if (variant == 0 || variant == 2) variant = splat_fwd_choice(B, N, R);
    const int S = split_parts(variant), P = split_part_rays(N, S);
    const long part_bytes = split_partial_bytes(B, R, S);
    if (S > 1 && (!scratch || scratch_bytes < part_bytes)) return HELIO_E_SCRATCH;
    CullFwd cull{nullptr, nullptr, nullptr};
    if (const CullPlan c = scratch ? cull_fwd_plan(variant, B, N, R) : CullPlan{0, 1, 0, false};
        c.te && scratch_bytes >= part_bytes + cull_fwd_bytes(B, N, R, c.te, c.S, c.P))
        cull = launch_cull_fwd(B, N, R, c.te, c.S, c.P, c.order, rays, xs, ys, static_cast<char*>(scratch) + part_bytes, st);
    if (S > 1) {
        float* partials = static_cast<float*>(scratch);
        launch_tile<4, false>(B, N, R, rays, xs, ys, image, st, cull, S, P, partials);
        const long px = (long)R * R;
        hipLaunchKernelGGL(splat_reduce_parts, dim3((unsigned)min(4096l, (px + 255) / 256), B), dim3(256), 0, st, px, S,
                           partials, image);
        return HELIO_OK;
    }
    switch (variant) {
    case 9: {       // the k-split block kernel; forced: 16 waves where the rule would not choose it
        const int kp = ksplit_parts(B, N, R);
        return launch_ksplit(B, N, R, rays, xs, ys, image, kp ? kp : (N >= 512 ? 16 : 4), st, cull) ? HELIO_OK : HELIO_E_INVALID;
    }
    case 1:
        if (t128 < 512) {
            const int t = (R + 63) / 64;
            hipLaunchKernelGGL((splat_fwd_valu<64, 32>), dim3(t * t, B), dim3(256), 0, st, B, N, R, rays, xs, ys, image);
        } else {
            const int t = (R + 127) / 128;
            // one-level sums for the 8×8 register tile: 107 VGPRs instead of 224 (4 waves/SIMD instead
            // of 2) — 96 vs 76 TFLOP/s at config 4; the summation error stays within the tolerance
            hipLaunchKernelGGL((splat_fwd_valu<128, 32, false>), dim3(t * t, B), dim3(256), 0, st, B, N, R, rays, xs, ys, image);
        }
        return HELIO_OK;
    // one-level sums everywhere the heliostat loop is long (fewer registers → more waves per SIMD;
    // measured +4 %); the summation error at N = 5000 stays inside the tolerance (GPU tests)
    case 3: launch_regs<2, 2, 2, 2, 128, false>(B, N, R, rays, xs, ys, image, st, cull); return HELIO_OK;
    case 4: launch_tile<2, false>(B, N, R, rays, xs, ys, image, st, cull); return HELIO_OK;
    case 5: launch_tile<4, false>(B, N, R, rays, xs, ys, image, st, cull); return HELIO_OK;
    case 6: launch_regs<1, 1, 2, 2, 128, true>(B, N, R, rays, xs, ys, image, st); return HELIO_OK;
    case 7: launch_bf16x3<true>(B, N, R, rays, xs, ys, image, st); return HELIO_OK;
    case 8: launch_bf16x3<false>(B, N, R, rays, xs, ys, image, st); return HELIO_OK;
    default: return HELIO_E_INVALID;
    }
}

}  // namespace helio
