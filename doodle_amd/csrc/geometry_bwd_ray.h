// The per-ray backward of the geometry stage, shared by geometry_bwd_kernel (geometry.hip) and the fused
// small-problem backward (splat_bwd.hip: render_bwd_fused_small).  Adjoint of ray_trace.h's trace() given
// the five footprint moments of the ray, the cotangents of `actual` / `refl`, and — HelioEnv.step — the
// adjoint of the two ray losses.
#pragma once
#include <hip/hip_runtime.h>
#include "helio.h"
#include "helio_math.h"
#include "ray_trace.h"
#include "step_loss_math.h"

namespace helio {

// Adjoint of y = v / max(|v|, 1e-9) given n = the clamped norm and y.
__device__ __forceinline__ vec3 unit_bwd(vec3 gy, vec3 y, float n, bool clamped) {
    if (clamped) return {gy.x / n, gy.y / n, gy.z / n};          // norm path has zero gradient
    float p = gy.x * y.x + gy.y * y.y + gy.z * y.z;
    return {(gy.x - y.x * p) / n, (gy.y - y.y * p) / n, (gy.z - y.z * p) / n};
}

// what a ray's backward reads besides the moments; load() requests everything BEFORE the trace (a load placed
// behind it pays its first-touch latency — ≈900 cycles in a freshly launched kernel — a second time)
struct RayBwdIn {
    float4 tg; vec3 h, v, s, gr, ga_in;
    __device__ __forceinline__ void load(long m, int b, int n, const float* __restrict__ helios,
                                         const float* __restrict__ sun, const float* __restrict__ action,
                                         const float* __restrict__ trig, long trig_b_stride,
                                         const float* __restrict__ g_actual, const float* __restrict__ g_refl) {
        tg = *reinterpret_cast<const float4*>(trig + (long)b * trig_b_stride + 4l * n);
        h = ld3(helios + 3l * n);
        gr = g_refl ? ld3(g_refl + 3 * m) : vec3{0.f, 0.f, 0.f};
        ga_in = g_actual ? ld3(g_actual + 3 * m) : vec3{0.f, 0.f, 0.f};
        v = ld3(action + 3 * m);
        s = ld3(sun + 3l * b);
    }
};

// → d loss / d action[m]; have_moments == false: no gradient arrives through the image
__device__ __forceinline__ vec3 geometry_bwd_ray(const RayBwdIn& in, long m, int n, int B, int N, const PlaneK& P,
                                                 bool have_moments, float M0, float Mx, float My, float Mxx, float Myy,
                                                 bool have_g_actual, const float* __restrict__ helios,
                                                 const float* __restrict__ action, const RayLossBwdArgs& RL) {
    const float LN2 = 0.69314718055994530942f;
    const float ce = in.tg.x, se = in.tg.y, cu = in.tg.z, su = in.tg.w;
    const vec3 h = in.h, ga_in = in.ga_in;
    vec3 gr = in.gr;
    Ray q = trace(in.v, ce, se, cu, su, h, in.s, P);

    if (have_moments && q.valid) {
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
    if (have_g_actual) { gact.x += ga_in.x; gact.y += ga_in.y; gact.z += ga_in.z; }
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
    if (RL.ideal && RL.g_bound) return {gxin + lv[0], gyin + lv[1], gz + lv[2]};
    return {gxin, gyin, gz};

}

}  // namespace helio
