// Per-ray and per-block pieces of HelioEnv.step's loss block, shared by step_losses.hip and the
// fused small-problem env kernel in splat_fwd.hip.  Reference: test_environment.py:101-155.
#pragma once
#include <hip/hip_runtime.h>

namespace helio {

struct LossGeom {                        // host constants of boundary(), by value
    float tp[3], tn[3];                  // target position / normal (as the env stores them)
    float hw, hh;                        // 0.75·W/2, 0.75·H/2   (test_environment.py:123)
    float hwt, hht;                      // hw·0.75, hh·0.75     (:124)
    int exponential_risk;
};

__device__ __forceinline__ float block_sum(float v, float* scratch) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// per-ray forward of the two ray losses; shared by forward and backward
struct RayLoss {
    float c, ang;            // clamped cosine, angle [mrad]
    bool clamped;
    float out;               // boundary term
    // intermediates for the adjoint
    float t, den, xl, yl, dx, dy, dist;
    bool inside;
};

__device__ __forceinline__ RayLoss ray_loss(const float* __restrict__ ideal, const float* __restrict__ actual,
                                            const float* __restrict__ v, const float* __restrict__ h,
                                            const LossGeom& g) {
    RayLoss r;
    // :144-155  acos(clamp(<ideal,actual>)) · 1000
    const float c0 = (ideal[0] * actual[0] + ideal[1] * actual[1]) + ideal[2] * actual[2];
    const float hi = 0.99999994f;        // nextafter(1,0) - 1e-10, rounded to fp32
    r.clamped = !(c0 > -hi && c0 < hi);
    r.c = c0 != c0 ? c0 : fminf(fmaxf(c0, -hi), hi);          // torch.clamp keeps a NaN
    r.ang = acosf(r.c) * 1000.0f;
    // :115-130  boundary()
    const float dots = -((v[0] * g.tn[0] + v[1] * g.tn[1]) + v[2] * g.tn[2]);
    const bool valid = fabsf(dots) > 1e-6f;
    r.den = dots + (valid ? 0.0f : 1e-6f);
    r.t = ((g.tp[0] * v[0] + g.tp[1] * v[1]) + g.tp[2] * v[2]) / r.den;
    r.xl = (h[0] + v[0] * r.t) - g.tp[0];            // local·(1,0,0)
    r.yl = (h[2] + v[2] * r.t) - g.tp[2];            // local·(0,0,1)
    // F.relu keeps a NaN (fmaxf would drop it): a NaN normal must poison the boundary loss, which
    // the reference's asserts then report (:497, :501)
    const float ex = fabsf(r.xl) - g.hwt, ey = fabsf(r.yl) - g.hht;
    r.dx = (ex > 0.0f || ex != ex) ? ex : 0.0f;
    r.dy = (ey > 0.0f || ey != ey) ? ey : 0.0f;
    r.dist = sqrtf((r.dx * r.dx + r.dy * r.dy) + 1e-8f);
    r.inside = fabsf(r.xl) <= g.hw && fabsf(r.yl) <= g.hh && valid;
    r.out = r.inside ? 0.0f : r.dist;
    return r;
}

inline LossGeom make_geom(const float* tp, const float* tn, float W, float H, int exponential_risk) {
    LossGeom g;
    for (int k = 0; k < 3; ++k) { g.tp[k] = tp[k]; g.tn[k] = tn[k]; }
    g.hw = (W * 0.75f) / 2.0f; g.hh = (H * 0.75f) / 2.0f;
    g.hwt = g.hw * 0.75f; g.hht = g.hh * 0.75f;
    g.exponential_risk = exponential_risk;
    return g;
}

// Adjoint of ray_loss: ka = cotangent of the mean alignment angle / (B·N), go = cotangent of the
// mean boundary term / (B·N).  ga[3] → d/d actual (through the angle), gv[3] → d/d action
// (through the boundary term); either may be skipped with a zero factor.
__device__ __forceinline__ void ray_loss_bwd(const RayLoss& r, const float* __restrict__ ideal,
                                             const float* __restrict__ v, const LossGeom& g, float ka, float go,
                                             float ga[3], float gv[3]) {
    // d acos(c)·1000 / dc = -1000 / sqrt(1 - c²); zero where the clamp is active
    const float k = r.clamped ? 0.0f : ka * (-1000.0f / sqrtf(1.0f - r.c * r.c));
    ga[0] = k * ideal[0]; ga[1] = k * ideal[1]; ga[2] = k * ideal[2];
    if (g.exponential_risk) go *= expf(r.out + 1e-6f);
    gv[0] = gv[1] = gv[2] = 0.0f;
    if (!r.inside) {
        const float gdx = go * r.dx / r.dist, gdy = go * r.dy / r.dist;
        const float gxl = (fabsf(r.xl) - g.hwt > 0.0f) ? (r.xl > 0.0f ? gdx : (r.xl < 0.0f ? -gdx : 0.0f)) : 0.0f;
        const float gyl = (fabsf(r.yl) - g.hht > 0.0f) ? (r.yl > 0.0f ? gdy : (r.yl < 0.0f ? -gdy : 0.0f)) : 0.0f;
        // xl = h.x + v.x t - tp.x ; yl = h.z + v.z t - tp.z
        const float gt = gxl * v[0] + gyl * v[2];
        gv[0] = gxl * r.t; gv[2] = gyl * r.t;
        // t = (tp·v) / den ; den = -(v·tn) (+1e-6)
        // torch's div backward associates as grad·((num/den)/den); keep that order: with the
        // reference's target (position ∥ normal) t is a constant and the two terms below cancel
        const float gnum = gt / r.den, gden = -(gt * (r.t / r.den));
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2) gv[k2] += gnum * g.tp[k2] - gden * g.tn[k2];
    }
}

// the ray losses' adjoint folded into the geometry backward (helio_env_step_bwd): ideal == null → off
struct RayLossBwdArgs {
    const float* ideal; const float* g_align; const float* g_bound;    // cotangents: device scalars, may be null
    LossGeom g;
};

// d(mse, dist)/d img at one pixel (test_environment.py:436-457 differentiated): x = img, y = target,
// dm = distance map, s = target peak; km, kd carry the cotangents, the means and the error mask
__device__ __forceinline__ float loss_grad_pixel(float x, float y, float dm, float s, float km, float kd) {
    const float d = x / s - y / s;
    const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    return (km * d + kd * sg * dm) / s;
}

// per-image constants of loss_grad_pixel
struct LossGradArgs {
    const float* img; const float* target; const float* tx; const float* dmaps;
    const float* keep; const float* g_mse; const float* g_dist;     // keep / cotangents may be null
    __device__ __forceinline__ void constants(int b, int B, long P, float& s, float& km, float& kd) const {
        s = tx[b];
        const float kb = keep ? keep[b] : 1.0f;                                         // error mask (0/1)
        km = kb * (g_mse ? *g_mse : 0.0f) * 2.0f / ((float)B * (float)P);             // d mean(d²)
        kd = kb * (g_dist ? *g_dist : 0.0f) / (float)B;                                // d mean_b Σ e·dm
    }
};

// raw pointers of the loss block, by value into the fused env kernel
struct StepLossArgs {
    const float* target; const float* tx; const float* dmaps; const float* ideal;
    float* part_img; float* part_ray; float* align_err; float* all_bounds; float* aux;
    LossGeom g;
};

}  // namespace helio
