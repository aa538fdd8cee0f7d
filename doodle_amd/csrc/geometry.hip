// Per-ray geometry kernels (forward, backward, ideal normals) for gfx950.
//
// One thread per ray (b,n); a ray reads 12 B of action, 16 B of trig, 12 B of
// heliostat position (L2-resident) and writes 12+12+16 B: the stage is pure HBM
// streaming and a rounding error of the chip's time (the splat dominates).  What
// matters here is bit-fidelity with the reference (helio_math.h).
//
// Reference: newenv_rl_test_multi_error.py :78-104 (rotation), :369-373 (leaky
// ReLU + renormalise), :376-383 (incident, reflection), :52-75 (intersection),
// :126-127,:146 (sigma), :256-278 (ideal normals).
#include <hip/hip_runtime.h>
#include "helio.h"
#include "helio_math.h"
#include "ray_trace.h"
#include "step_loss_math.h"
#include "geometry_bwd_ray.h"

namespace helio {

__global__ void __launch_bounds__(256)
geometry_fwd_kernel(int B, int N, const float* __restrict__ helios, const float* __restrict__ sun,
                    const float* __restrict__ action, const float* __restrict__ trig, long trig_b_stride,
                    PlaneK P, float* __restrict__ actual, float* __restrict__ refl,
                    float* __restrict__ rays) {
    const long M = (long)B * N;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const int b = (int)(m / N), n = (int)(m - (long)b * N);
        const float4 tg = *reinterpret_cast<const float4*>(trig + (long)b * trig_b_stride + 4l * n);
        Ray q = trace(ld3(action + 3 * m), tg.x, tg.y, tg.z, tg.w, ld3(helios + 3l * n), ld3(sun + 3l * b), P);
        st3(actual + 3 * m, q.act);
        if (refl) st3(refl + 3 * m, q.r);
        if (rays) *reinterpret_cast<float4*>(rays + 4 * m) = make_float4(q.a, q.b, q.k2, q.c2);
    }
}

__global__ void __launch_bounds__(256)
geometry_bwd_kernel(int B, int N, int n_blocks,
                    const float* __restrict__ helios, const float* __restrict__ sun,
                    const float* __restrict__ action, const float* __restrict__ trig, long trig_b_stride,
                    PlaneK P, const float* __restrict__ moments,
                    const float* __restrict__ g_actual, const float* __restrict__ g_refl,
                    float* __restrict__ g_action, RayLossBwdArgs RL) {
    const long M = (long)B * N;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const int b = (int)(m / N), n = (int)(m - (long)b * N);
        RayBwdIn in;
        in.load(m, b, n, helios, sun, action, trig, trig_b_stride, g_actual, g_refl);
        // fixed-order sum of the column-block partials → deterministic
        float M0 = 0.f, Mx = 0.f, My = 0.f, Mxx = 0.f, Myy = 0.f;
        if (moments)
            for (int jb = 0; jb < n_blocks; ++jb) {
                const float* p = moments + (((long)b * n_blocks + jb) * N + n) * HELIO_MOMENT_STRIDE;
                M0 += p[0]; Mx += p[1]; My += p[2]; Mxx += p[3]; Myy += p[4];
            }
        st3(g_action + 3 * m, geometry_bwd_ray(in, m, n, B, N, P, moments != nullptr, M0, Mx, My, Mxx, Myy,
                                               g_actual != nullptr, helios, action, RL));
    }
}

__global__ void __launch_bounds__(256)
ideal_normals_kernel(int B, int N, const float* __restrict__ helios, const float* __restrict__ sun,
                     vec3 target, float* __restrict__ out) {
    const long M = (long)B * N;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const int b = (int)(m / N), n = (int)(m - (long)b * N);
        const vec3 h = ld3(helios + 3l * n);
        vec3 s = add3(unit3(sub3(ld3(sun + 3l * b), h)), unit3(sub3(target, h)));   // :264-266 / :275-277
        st3(out + 3 * m, unit3(s));                                                  // :267 / :278
    }
}

// init_actions, :293-303: noisy = ideal + randn_like(ideal) * noise (multiply, then add, each rounded),
// then noisy / norm(noisy).clamp_min(1e-9)
__global__ void __launch_bounds__(256)
init_actions_kernel(long M, const float* ideal, const float* noise, float scale, float* out) {
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const vec3 v = add3(ld3(ideal + 3 * m), scale3(scale, ld3(noise + 3 * m)));
        st3(out + 3 * m, unit3(v));
    }
}

// (cos_e, sin_e, cos_u, sin_u) of the error angles, :87-91: angle = err_mrad * 1e-3 (fp32), then
// the precise ocml sinf/cosf — the same device functions torch's own cos/sin kernels call.
__global__ void __launch_bounds__(256)
error_trig_kernel(long M, const float* __restrict__ errs, float* __restrict__ trig) {
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const float2 e = *reinterpret_cast<const float2*>(errs + 2 * m);
        const float ae = e.x * 1e-3f, au = e.y * 1e-3f;
        *reinterpret_cast<float4*>(trig + 4 * m) = make_float4(cosf(ae), sinf(ae), cosf(au), sinf(au));
    }
}

static inline int ray_grid(long M) {
    long g = (M + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

void launch_geometry_fwd(int B, int N, const float* helios, const float* sun, const float* action,
                         const float* trig, long trig_b_stride, const helio_plane* plane,
                         float* actual, float* refl, float* rays, hipStream_t st) {
    hipLaunchKernelGGL(geometry_fwd_kernel, dim3(ray_grid((long)B * N)), dim3(256), 0, st,
                       B, N, helios, sun, action, trig, trig_b_stride, to_k(plane), actual, refl, rays);
}

void launch_geometry_bwd(int B, int N, int n_blocks, const float* helios, const float* sun,
                         const float* action, const float* trig, long trig_b_stride,
                         const helio_plane* plane, const float* moments, const float* g_actual,
                         const float* g_refl, float* g_action, hipStream_t st) {
    hipLaunchKernelGGL(geometry_bwd_kernel, dim3(ray_grid((long)B * N)), dim3(256), 0, st,
                       B, N, n_blocks, helios, sun, action, trig, trig_b_stride, to_k(plane), moments,
                       g_actual, g_refl, g_action, RayLossBwdArgs{});
}

// geometry backward with the adjoint of HelioEnv.step's two ray losses folded in
void launch_geometry_bwd_losses(int B, int N, int n_blocks, const float* helios, const float* sun,
                                const float* action, const float* trig, long trig_b_stride,
                                const helio_plane* plane, const float* moments, const float* g_actual,
                                const float* g_refl, float* g_action, const float* ideal, const float* g_align,
                                const float* g_bound, const float* tp, const float* tn, float W, float H,
                                int exponential_risk, hipStream_t st) {
    RayLossBwdArgs RL;
    RL.ideal = ideal; RL.g_align = g_align; RL.g_bound = g_bound;
    RL.g = make_geom(tp, tn, W, H, exponential_risk);
    hipLaunchKernelGGL(geometry_bwd_kernel, dim3(ray_grid((long)B * N)), dim3(256), 0, st,
                       B, N, n_blocks, helios, sun, action, trig, trig_b_stride, to_k(plane), moments,
                       g_actual, g_refl, g_action, RL);
}

void launch_error_trig(long M, const float* errs, float* trig, hipStream_t st) {
    hipLaunchKernelGGL(error_trig_kernel, dim3(ray_grid(M)), dim3(256), 0, st, M, errs, trig);
}

void launch_init_actions(long M, const float* ideal, const float* noise, float scale, float* out, hipStream_t st) {
    hipLaunchKernelGGL(init_actions_kernel, dim3(ray_grid(M)), dim3(256), 0, st, M, ideal, noise, scale, out);
}

void launch_ideal_normals(int B, int N, const float* helios, const float* sun, const float* target,
                          float* out, hipStream_t st) {
    hipLaunchKernelGGL(ideal_normals_kernel, dim3(ray_grid((long)B * N)), dim3(256), 0, st,
                       B, N, helios, sun, vec3{target[0], target[1], target[2]}, out);
}

}  // namespace helio
