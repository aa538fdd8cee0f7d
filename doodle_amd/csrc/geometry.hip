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

// Adjoint of y = v / max(|v|, 1e-9) given n = the clamped norm and y.
__device__ __forceinline__ vec3 unit_bwd(vec3 gy, vec3 y, float n, bool clamped) {
    if (clamped) return {gy.x / n, gy.y / n, gy.z / n};          // norm path has zero gradient
    float p = gy.x * y.x + gy.y * y.y + gy.z * y.z;
    return {(gy.x - y.x * p) / n, (gy.y - y.y * p) / n, (gy.z - y.z * p) / n};
}

__global__ void __launch_bounds__(256)
geometry_bwd_kernel(int B, int N, int n_blocks,
                    const float* __restrict__ helios, const float* __restrict__ sun,
                    const float* __restrict__ action, const float* __restrict__ trig, long trig_b_stride,
                    PlaneK P, const float* __restrict__ moments,
                    const float* __restrict__ g_actual, const float* __restrict__ g_refl,
                    float* __restrict__ g_action, RayLossBwdArgs RL) {
    const float LN2 = 0.69314718055994530942f;
    const long M = (long)B * N;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const int b = (int)(m / N), n = (int)(m - (long)b * N);
        const float4 tg = *reinterpret_cast<const float4*>(trig + (long)b * trig_b_stride + 4l * n);
        const float ce = tg.x, se = tg.y, cu = tg.z, su = tg.w;
        const vec3 h = ld3(helios + 3l * n);
        // every input of the ray is requested BEFORE the trace: a load placed behind it (the moments were)
        // pays its first-touch latency (≈900 cycles in a freshly launched kernel) a second time
        vec3 gr = g_refl ? ld3(g_refl + 3 * m) : vec3{0.f, 0.f, 0.f};
        const vec3 ga_in = g_actual ? ld3(g_actual + 3 * m) : vec3{0.f, 0.f, 0.f};
        // fixed-order sum of the column-block partials → deterministic
        float M0 = 0.f, Mx = 0.f, My = 0.f, Mxx = 0.f, Myy = 0.f;
        if (moments)
            for (int jb = 0; jb < n_blocks; ++jb) {
                const float* p = moments + (((long)b * n_blocks + jb) * N + n) * HELIO_MOMENT_STRIDE;
                M0 += p[0]; Mx += p[1]; My += p[2]; Mxx += p[3]; Myy += p[4];
            }
        Ray q = trace(ld3(action + 3 * m), ce, se, cu, su, h, ld3(sun + 3l * b), P);

        if (moments && q.valid) {
            // gauss = exp2(-q k2), q = t² + s² + c2  →  cotangents of (a, b, k2, c2)
            const float ga = -2.0f * LN2 * q.k2 * Mx;
            const float gb = -2.0f * LN2 * q.k2 * My;
            const float gk2 = -LN2 * (Mxx + Myy + q.c2 * M0);
            const float gc2 = -LN2 * q.k2 * M0;
            // a = d0·u, b = d0·v, c2 = (d0·w)², d0 = o - x
            const float gc = 2.0f * q.c * gc2;
            vec3 gx = {-(ga * P.u.x + gb * P.v.x + gc * P.w.x),
                       -(ga * P.u.y + gb * P.v.y + gc * P.w.y),
                       -(ga * P.u.z + gb * P.v.z + gc * P.w.z)};
            // k2 = log2e / max(2σ²,1e-12); σ = max(σs |x-h|, 1e-9)
            if (q.two_raw >= 1e-12f && q.sraw >= 1e-9f && q.dist > 0.0f) {
                const float g_two = -gk2 * q.k2 / q.two_s2;
                const float g_dist = g_two * 4.0f * q.sigma * P.sigma_scale;
                gx.x += g_dist * q.dh.x / q.dist;
                gx.y += g_dist * q.dh.y / q.dist;
                gx.z += g_dist * q.dh.z / q.dist;
            }
            // x = h + t r,  t = num / denom,  denom = r·p̂
            const float gt = gx.x * q.r.x + gx.y * q.r.y + gx.z * q.r.z;
            const float gden = -gt * q.t / q.denom;
            gr.x += q.t * gx.x + gden * q.phat.x;
            gr.y += q.t * gx.y + gden * q.phat.y;
            gr.z += q.t * gx.z + gden * q.phat.z;
        }
        // r = r0 / max(|r0|,1e-9)
        vec3 gr0 = unit_bwd(gr, q.r, q.nr0, norm3(q.r0) < 1e-9f);
        // r0 = -inc - (2 dots) n̂ ;  dots = -(inc·n̂)   (inc does not depend on the action)
        const float gdots = -2.0f * (gr0.x * q.nh.x + gr0.y * q.nh.y + gr0.z * q.nh.z);
        const float two = 2.0f * q.dots;
        vec3 gnh = {-two * gr0.x - gdots * q.inc.x, -two * gr0.y - gdots * q.inc.y, -two * gr0.z - gdots * q.inc.z};
        // n̂ = act / max(|act|,1e-9) ; act also is an output
        vec3 gact = unit_bwd(gnh, q.nh, q.na, norm3(q.act) < 1e-9f);
        if (g_actual) { gact.x += ga_in.x; gact.y += ga_in.y; gact.z += ga_in.z; }
        // HelioEnv.step's ray losses (alignment angle of `actual`, boundary term of the action itself)
        float lv[3] = {0.f, 0.f, 0.f};
        if (RL.ideal) {
            const float act[3] = {q.act.x, q.act.y, q.act.z};
            const RayLoss r = ray_loss(RL.ideal + 3 * m, act, action + 3 * m, helios + 3l * n, RL.g);
            const float inv = 1.0f / ((float)B * (float)N);
            float la[3];
            ray_loss_bwd(r, RL.ideal + 3 * m, action + 3 * m, RL.g, (RL.g_align ? *RL.g_align : 0.0f) * inv,
                         (RL.g_bound ? *RL.g_bound : 0.0f) * inv, la, lv);
            if (RL.g_align) { gact.x += la[0]; gact.y += la[1]; gact.z += la[2]; }
            if (!RL.g_bound) lv[0] = lv[1] = lv[2] = 0.0f;
        }
        // act = vrot / max(|vrot|,1e-9)
        vec3 gv = unit_bwd(gact, q.act, q.nv, norm3(q.vrot) < 1e-9f);
        // leaky ReLU on Z, then the two rotations transposed
        const float gze = q.ze_pre > 0.0f ? gv.z : gv.z * 0.01f;
        const float gyu = ce * gv.y + se * gze;
        const float gz = -se * gv.y + ce * gze;
        const float gxin = cu * gv.x + su * gyu;
        const float gyin = -su * gv.x + cu * gyu;
        if (RL.ideal && RL.g_bound) st3(g_action + 3 * m, {gxin + lv[0], gyin + lv[1], gz + lv[2]});
        else st3(g_action + 3 * m, {gxin, gyin, gz});
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
