// Exact Euclidean distance maps on the device (SURVEY.md §8 f, row 3).
//
// Reference: make_distance_maps, test_environment.py:92-97 — per image,
//   mask = img > thr·max(img);  map = scipy.ndimage.distance_transform_edt(1 - mask)
// i.e. the Euclidean distance (in pixels) of every pixel to the nearest pixel ABOVE the
// threshold.  The reference round-trips through the host (imgs.cpu().numpy() → scipy → back);
// here it stays on the device: separable exact EDT, integer arithmetic, one fp64 sqrt.
//   pass 0  per image: max (fixed-order tree reduction)
//   pass 1  per column: vertical distance g[i][j] to the nearest hot pixel of the column
//   pass 2  per pixel : d² = min_j' (j-j')² + g[i][j']²   (row of g in LDS)
// Degenerate image with no hot pixel: scipy measures from a virtual background pixel at
// (-1, 0); reproduced.  Runs only in HelioEnv.set_sun_pos, so clarity beats tuning.
#include <hip/hip_runtime.h>
#include "helio.h"

namespace helio {

constexpr int EDT_INF = 1 << 20;

__global__ void __launch_bounds__(256)
edt_max_kernel(int R, const float* __restrict__ img, float* __restrict__ mx, int* __restrict__ any_hot) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = img + (long)b * R * R;
    float m = -INFINITY;
    for (long k = tid; k < (long)R * R; k += 256) m = fmaxf(m, p[k]);
    red[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (tid == 0) { mx[b] = red[0]; any_hot[b] = 0; }
}

__global__ void __launch_bounds__(256)
edt_columns_kernel(int B, int R, const float* __restrict__ img, const float* __restrict__ mx, float thr,
                   int* __restrict__ g, int* __restrict__ any_hot) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;     // (b, column)
    if (c >= (long)B * R) return;
    const int b = (int)(c / R), j = (int)(c % R);
    const float* p = img + (long)b * R * R + j;
    int* q = g + (long)b * R * R + j;
    const float cut = thr * mx[b];
    int d = EDT_INF, hot = 0;
    for (int i = 0; i < R; ++i) {                            // nearest hot pixel above
        const bool h = p[(long)i * R] > cut;
        d = h ? 0 : (d >= EDT_INF ? EDT_INF : d + 1);
        hot |= h;
        q[(long)i * R] = d;
    }
    d = EDT_INF;
    for (int i = R - 1; i >= 0; --i) {                       // nearest hot pixel below
        const int up = q[(long)i * R];
        d = up == 0 ? 0 : (d >= EDT_INF ? EDT_INF : d + 1);
        q[(long)i * R] = min(up, d);
    }
    if (hot) atomicOr(any_hot + b, 1);
}

__global__ void __launch_bounds__(256)
edt_rows_kernel(int R, const int* __restrict__ g, const int* __restrict__ any_hot, float* __restrict__ out) {
    extern __shared__ int row[];
    const int b = blockIdx.y, i = blockIdx.x;
    const int* gr = g + ((long)b * R + i) * R;
    for (int j = threadIdx.x; j < R; j += 256) row[j] = gr[j];
    __syncthreads();
    const bool degenerate = any_hot[b] == 0;
    for (int j = threadIdx.x; j < R; j += 256) {
        long long best;
        if (degenerate) {
            best = (long long)(i + 1) * (i + 1) + (long long)j * j;
        } else {
            best = (long long)EDT_INF * EDT_INF;
            for (int k = 0; k < R; ++k) {
                const long long v = row[k], dj = j - k;
                const long long d2 = v * v + dj * dj;
                best = d2 < best ? d2 : best;
            }
        }
        out[((long)b * R + i) * R + j] = (float)sqrt((double)best);
    }
}

void launch_distance_maps(int B, int R, const float* img, float thr, int* ws_g, float* ws_mx, int* ws_any,
                          float* out, hipStream_t st) {
    hipLaunchKernelGGL(edt_max_kernel, dim3(B), dim3(256), 0, st, R, img, ws_mx, ws_any);
    hipLaunchKernelGGL(edt_columns_kernel, dim3((int)(((long)B * R + 255) / 256)), dim3(256), 0, st, B, R, img, ws_mx,
                       thr, ws_g, ws_any);
    hipLaunchKernelGGL(edt_rows_kernel, dim3(R, B), dim3(256), R * sizeof(int), st, R, ws_g, ws_any, out);
}

}  // namespace helio
