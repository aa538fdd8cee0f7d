// Order-preserving compaction of the rays that are not exactly zero on a tile (forward) or on an image
// (backward) — see cull.h for why leaving the others out is bit-identical.  One workgroup per list;
// deterministic (ballot + prefix counts, no atomics): a list keeps the heliostat order, so the footprint
// kernels add the surviving terms in the order the dense kernels add them.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "helio.h"
#include "cull.h"

namespace helio {

constexpr int CULL_THREADS = 256;
constexpr int CULL_BWD_THREADS = 1024;       // one workgroup per image compacts its list: few images must not take long

// min and max of v[i0 .. i1) over the workgroup (every thread gets both); i1 > i0
template <int THREADS = CULL_THREADS>
__device__ __forceinline__ void block_minmax(const float* __restrict__ v, int i0, int i1, float* sm, float& lo, float& hi) {
    constexpr int CULL_WAVES = THREADS / 64;
    float a = __builtin_inff(), b = -__builtin_inff();
    bool bad = false;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += THREADS) {
        const float x = v[i];
        bad |= x != x;
        a = fminf(a, x);
        b = fmaxf(b, x);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a = fminf(a, __shfl_xor(a, d)); b = fmaxf(b, __shfl_xor(b, d)); }
    const bool any_bad = __any(bad);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[3 * wave] = a; sm[3 * wave + 1] = b; sm[3 * wave + 2] = any_bad ? 1.0f : 0.0f; }
    __syncthreads();
    lo = sm[0]; hi = sm[1];
    float nb = sm[2];
#pragma unroll
    for (int w = 1; w < CULL_WAVES; ++w) { lo = fminf(lo, sm[3 * w]); hi = fmaxf(hi, sm[3 * w + 1]); nb += sm[3 * w + 2]; }
    if (nb != 0.0f) lo = hi = __builtin_nanf("");      // a NaN coordinate: every ray of the tile is kept
    __syncthreads();
}

// the calling thread's place among the flagged threads of the workgroup, and how many there are
template <int THREADS = CULL_THREADS>
__device__ __forceinline__ int block_rank(bool flag, int* sw, int& total) {
    constexpr int CULL_WAVES = THREADS / 64;
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sw[wave] = __popcll(m);
    __syncthreads();
    int before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < CULL_WAVES; ++w) { const int c = sw[w]; before += w < wave ? c : 0; total += c; }
    __syncthreads();
    return before + rank;
}

// grid (tiles²·S, B): the rays [part·P, part·P + P) of image b that are not exactly zero on tile (ti, tj) —
// blockIdx.x = (ti·tiles + tj)·S + part, the tile (and part) numbering of the footprint kernels; S = 1, P = N:
// the whole image's rays
__global__ void __launch_bounds__(CULL_THREADS)
cull_fwd_kernel(int Nall, int R, int TE, int S, int P, const float4* __restrict__ rays, const float* __restrict__ xs,
                const float* __restrict__ ys, int* __restrict__ counts, float4* __restrict__ lists) {
    __shared__ float sm[3 * CULL_THREADS / 64];
    __shared__ int sw[CULL_THREADS / 64];
    const int tiles = (R + TE - 1) / TE;
    const int tile = blockIdx.x / S, part = blockIdx.x % S;
    const int b = blockIdx.y, ti = tile / tiles, tj = tile % tiles;
    CullBox bx;
    block_minmax(xs, ti * TE, min(R, ti * TE + TE), sm, bx.xlo, bx.xhi);
    block_minmax(ys, tj * TE, min(R, tj * TE + TE), sm, bx.ylo, bx.yhi);
    const long list = (long)b * gridDim.x + blockIdx.x;
    const int first = part * P, N = max(0, min(Nall - first, P));
    const float4* __restrict__ rb = rays + (long)b * Nall + first;
    float4* __restrict__ out = lists + list * P;
    int base = 0;
    for (int n0 = 0; n0 < N; n0 += CULL_THREADS) {
        const int n = n0 + (int)threadIdx.x;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        bool live = false;
        if (n < N) { q = rb[n]; live = !cull_dead_product(q, bx); }
        int total;
        const int at = block_rank(live, sw, total);
        if (live) out[base + at] = q;
        base += total;
    }
    if (threadIdx.x == 0) counts[list] = base;
}

// The box a (pass, c tile) list is decided on: the c axis (columns in pass 0, rows in pass 1) restricted to the
// tile's TC pixels, the contracted axis whole.  CT = 1: the whole image on both axes (one list for both passes).
template <int THREADS>
__device__ __forceinline__ CullBox bwd_box(int R, int TC, int CT, int pass, int tile, const float* __restrict__ xs,
                                           const float* __restrict__ ys, float* sm) {
    const int c0 = CT > 1 ? tile * TC : 0, c1 = CT > 1 ? min(R, c0 + TC) : R;
    CullBox bx;
    block_minmax<THREADS>(xs, pass == 1 ? c0 : 0, pass == 1 ? c1 : R, sm, bx.xlo, bx.xhi);
    block_minmax<THREADS>(ys, pass == 0 ? c0 : 0, pass == 0 ? c1 : R, sm, bx.ylo, bx.yhi);
    return bx;
}

// grid (CT, B, sets): the rays of image b whose moments of this (pass, c tile) are not identically zero, as
// indices, in order; list = (pass·B + b)·CT + tile.  THREADS = 1024 with few lists (few images must not take
// long), 256 with many (all of them resident at once: 51 → 20 µs at config 4).
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
cull_bwd_kernel(int N, int R, int TC, int CT, const float4* __restrict__ rays, const float* __restrict__ xs,
                const float* __restrict__ ys, int* __restrict__ counts, int* __restrict__ idx) {
    __shared__ float sm[3 * THREADS / 64];
    __shared__ int sw[THREADS / 64];
    const int tile = blockIdx.x, b = blockIdx.y, pass = blockIdx.z;
    const long list = ((long)pass * gridDim.y + b) * CT + tile;
    const CullBox bx = bwd_box<THREADS>(R, TC, CT, pass, tile, xs, ys, sm);
    const float4* __restrict__ rb = rays + (long)b * N;
    int* __restrict__ out = idx + list * N;
    int base = 0;
    for (int n0 = 0; n0 < N; n0 += THREADS) {
        const int n = n0 + (int)threadIdx.x;
        const bool live = n < N && !cull_dead_strict(rb[n], bx);
        int total;
        const int at = block_rank<THREADS>(live, sw, total);
        if (live) out[base + at] = n;
        base += total;
    }
    if (threadIdx.x == 0) counts[list] = base;
}

// The moments a ray is NOT listed for must read zero — what the dense kernels compute for them (pass 0 owns
// components (0, 2, 4) of a tile's 64-wide column blocks, pass 1 components (1, 3) of its row blocks).  The whole
// buffer is cleared in front of the passes, which then write the listed rays' entries: whole cache lines at
// streaming rate.  (A kernel that zeroed exactly the dead entries — the criterion once more, per ray — took 81 µs
// at config 4, 54 µs with coalesced stores: partial lines; this takes what 164 MB of stores take.)
__global__ void __launch_bounds__(CULL_THREADS)
cull_zero_kernel(float* __restrict__ p, long n) {
    // 16-byte stores over the aligned middle, dwords at the ragged ends (the ABI asks 4-byte alignment of moments_d)
    const long head = min(n, (long)((16 - (reinterpret_cast<unsigned long long>(p) & 15)) & 15) / 4);
    const long n4 = (n - head) / 4;
    const long tid = blockIdx.x * (long)CULL_THREADS + threadIdx.x, stride = (long)gridDim.x * CULL_THREADS;
    float4* __restrict__ q = reinterpret_cast<float4*>(p + head);
    for (long i = tid; i < n4; i += stride) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < head) p[tid] = 0.0f;
    const long tail0 = head + 4 * n4;
    if (tid < n - tail0) p[tail0 + tid] = 0.0f;
}

// ---- work order (cull.h): one workgroup each -----------------------------------------------------------
constexpr int ORDER_THREADS = 1024, ORDER_WAVES = ORDER_THREADS / 64;

// inclusive prefix sum of one int per thread over the workgroup; `total` = the sum (all threads)
__device__ __forceinline__ int block_scan_incl(int v, int* sw, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) sw[wave] = x;
    __syncthreads();
    int before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < ORDER_WAVES; ++w) { const int c = sw[w]; before += w < wave ? c : 0; total += c; }
    __syncthreads();
    return before + x;
}

// order[] = the T lists sorted by length, longest first: a counting sort on 1024 length classes (the order
// inside a class is whatever the atomics give — it decides which workgroup computes a tile, nothing else)
__global__ void __launch_bounds__(ORDER_THREADS)
cull_order_fwd_kernel(int T, int N, const int* __restrict__ counts, int* __restrict__ order) {
    __shared__ int cls[ORDER_THREADS];
    __shared__ int sw[ORDER_WAVES];
    const int tid = threadIdx.x;
    auto klass = [&](int c) { return 1023 - (int)min(1023l, (long)c * 1023 / max(N, 1)); };     // 0 = the longest
    cls[tid] = 0;
    __syncthreads();
    for (int i = tid; i < T; i += ORDER_THREADS) atomicAdd(&cls[klass(counts[i])], 1);
    __syncthreads();
    const int mine = cls[tid];
    int total;
    const int incl = block_scan_incl(mine, sw, total);
    cls[tid] = incl - mine;                                   // where the class starts
    __syncthreads();
    for (int i = tid; i < T; i += ORDER_THREADS) order[atomicAdd(&cls[klass(counts[i])], 1)] = i;
}

// map[] = (list, ray tile) of every tile of the lists that holds a ray, list-major (= image-major); *total = how
// many.  With split_tails the last tile of a list goes to tail_map[] instead when it holds at most 128 rays.
// grid (sets): one workgroup per set of `lists` lists.
__global__ void __launch_bounds__(ORDER_THREADS)
cull_map_bwd_kernel(int lists, int nt, int split_tails, int tile_rays, const int* __restrict__ counts, int* __restrict__ total_out,
                    int2* __restrict__ map, int* __restrict__ tail_total_out, int2* __restrict__ tail_map) {
    __shared__ int sw[ORDER_WAVES];
    counts += (long)blockIdx.x * lists;
    map += (long)blockIdx.x * lists * nt;
    tail_map += (long)blockIdx.x * lists;
    int running = 0, running_tail = 0;
    for (int l0 = 0; l0 < lists; l0 += ORDER_THREADS) {
        const int l = l0 + (int)threadIdx.x;
        const int c = l < lists ? counts[l] : 0;
        const int t = (c + tile_rays - 1) / tile_rays;
        const int tail = split_tails && t > 0 && c - (t - 1) * tile_rays <= tile_rays / 2;
        const int tf = t - tail;
        int total;
        const int start = running + block_scan_incl(tf, sw, total) - tf;
        for (int k = 0; k < tf; ++k) map[start + k] = make_int2(l, k);
        running += total;
        const int start_t = running_tail + block_scan_incl(tail, sw, total) - tail;
        if (tail) tail_map[start_t] = make_int2(l, t - 1);
        running_tail += total;
    }
    if (threadIdx.x == 0) { total_out[blockIdx.x] = running; tail_total_out[blockIdx.x] = running_tail; }
}

// HELIO_CULL=0 switches the stage off (A/B runs): the dense kernels then run whatever scratch is passed
bool cull_enabled() {
    static const bool on = [] { const char* e = getenv("HELIO_CULL"); return !(e && e[0] == '0'); }();
    return on;
}

CullFwd launch_cull_fwd(int B, int N, int R, int TE, int S, int P, bool with_order, const float* rays, const float* xs,
                        const float* ys, void* scratch, hipStream_t st) {
    const int t = (R + TE - 1) / TE;
    const long T = (long)B * t * t * S;
    if (S == 1) P = N;
    char* base = static_cast<char*>(scratch);
    int* counts = reinterpret_cast<int*>(base);
    int* order = reinterpret_cast<int*>(base + cull_pad256(4 * T));
    float4* lists = reinterpret_cast<float4*>(base + 2 * cull_pad256(4 * T));
    hipLaunchKernelGGL(cull_fwd_kernel, dim3(t * t * S, B), dim3(CULL_THREADS), 0, st, N, R, TE, S, P,
                       reinterpret_cast<const float4*>(rays), xs, ys, counts, lists);
    if (with_order)
        hipLaunchKernelGGL(cull_order_fwd_kernel, dim3(1), dim3(ORDER_THREADS), 0, st, (int)T, P, counts, order);
    return CullFwd{counts, with_order ? order : nullptr, lists};
}

CullBwd launch_cull_bwd(int B, int N, int R, int JB, int TC, int CT, bool with_map, bool split_tails, const float* rays, const float* xs,
                        const float* ys, float* moments, void* scratch, hipStream_t st, int tile_rays) {
    const int sets = CT > 1 ? 2 : 1;
    const long T = cull_bwd_lists(B, CT), nt = (N + tile_rays - 1) / tile_rays;
    char* base = static_cast<char*>(scratch);
    int* counts = reinterpret_cast<int*>(base);
    int* idx = reinterpret_cast<int*>(base + cull_pad256(4 * T));
    int* total = reinterpret_cast<int*>(base + cull_pad256(4 * T) + cull_pad256(4 * T * N));
    int2* map = reinterpret_cast<int2*>(reinterpret_cast<char*>(total) + 256);
    if (T >= 512)
        hipLaunchKernelGGL(cull_bwd_kernel<256>, dim3(CT, B, sets), dim3(256), 0, st, N, R, TC, CT,
                           reinterpret_cast<const float4*>(rays), xs, ys, counts, idx);
    else
        hipLaunchKernelGGL(cull_bwd_kernel<CULL_BWD_THREADS>, dim3(CT, B, sets), dim3(CULL_BWD_THREADS), 0, st, N, R, TC, CT,
                           reinterpret_cast<const float4*>(rays), xs, ys, counts, idx);
    const long nm = (long)B * JB * N * HELIO_MOMENT_STRIDE;
    hipLaunchKernelGGL(cull_zero_kernel, dim3((unsigned)min(8192l, (nm / 4 + CULL_THREADS - 1) / CULL_THREADS + 1)), dim3(CULL_THREADS), 0, st,
                       moments, nm);
    int* tail_total = total + 8;
    int2* tail_map = map + T * nt;
    if (with_map)
        hipLaunchKernelGGL(cull_map_bwd_kernel, dim3(sets), dim3(ORDER_THREADS), 0, st, (int)(T / sets), (int)nt, (int)split_tails, tile_rays, counts,
                           total, map, tail_total, tail_map);
    const bool tails = with_map && split_tails;
    return CullBwd{counts, idx, total, map, tails ? tail_total : nullptr, tails ? tail_map : nullptr, CT, T / sets, T / sets * nt, N};
}

}  // namespace helio
