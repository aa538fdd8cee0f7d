// Per-ray forward geometry shared by the geometry kernels and the fused small-problem
// render kernel.  Bit-faithful restatement of the reference (see helio_math.h).
//
// Reference: newenv_rl_test_multi_error.py :78-104 (rotation), :369-373 (leaky ReLU +
// renormalise), :376-383 (incident, reflection), :52-75 (intersection), :126-127,:146 (sigma).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include "helio.h"
#include "helio_math.h"

namespace helio {

struct PlaneK {  // helio_plane by value in kernel-argument space
    vec3 o, nrm, u, v, w;
    vec3 phat;       // nrm / max(|nrm|, 1e-9), :60 — the same for every ray: divided once, on the host (to_k)
    float sigma_scale;
};

// Everything the forward computes for one ray; the backward recomputes it.
struct Ray {
    float ze_pre;          // rotated Z before the leaky ReLU
    vec3 vrot;             // rotated + leaky-clamped, un-normalised
    float nv;              // max(|vrot|,1e-9)
    vec3 act;              // `actual` (output #2)
    vec3 inc;              // unit incident direction (heliostat → sun)
    float na;              // max(|act|,1e-9)
    vec3 nh;               // act re-normalised inside reflect_vectors
    float dots;
    vec3 r0;  float nr0;   // un-normalised reflection and its clamped norm
    vec3 r;                // `refl` (output #3)
    vec3 phat;
    float denom, t;  bool valid;
    vec3 x;                // intersection (0 if invalid)
    vec3 dh;  float dist, sraw, sigma, two_raw, two_s2;
    vec3 d0;  float a, b, c, c2, k2;
};

// trace() in two halves: everything up to the unit reflection needs no receiver-plane constant — the
// fused small-problem kernel runs that half while its plane constants are still on their way from the
// kernel-argument segment.  trace() = head + tail: the same operations in the same order.
__device__ __forceinline__ void trace_head(Ray& q, vec3 nin, float ce, float se, float cu, float su, vec3 h, vec3 s) {
    // :96-102  rotate about Up (Z) then East (X); every op individually rounded
    float xu = __fsub_rn(__fmul_rn(cu, nin.x), __fmul_rn(su, nin.y));
    float yu = __fadd_rn(__fmul_rn(su, nin.x), __fmul_rn(cu, nin.y));
    float ye = __fsub_rn(__fmul_rn(ce, yu), __fmul_rn(se, nin.z));
    float ze = __fadd_rn(__fmul_rn(se, yu), __fmul_rn(ce, nin.z));
    q.ze_pre = ze;
    ze = ze > 0.0f ? ze : __fmul_rn(ze, 0.01f);                  // :369 leaky_relu(0.01)
    q.vrot = {xu, ye, ze};
    q.act = unit3(q.vrot, q.nv);                                  // :372
    q.inc = unit3(sub3(s, h));                                    // :377-380
    q.nh = unit3(q.act, q.na);                                    // :48
    q.dots = -dot3(q.inc, q.nh);                                  // :49
    float two = __fmul_rn(2.0f, q.dots);
    q.r0 = {__fsub_rn(-q.inc.x, __fmul_rn(two, q.nh.x)),          // :50
            __fsub_rn(-q.inc.y, __fmul_rn(two, q.nh.y)),
            __fsub_rn(-q.inc.z, __fmul_rn(two, q.nh.z))};
    q.r = unit3(q.r0, q.nr0);                                     // :383
}

__device__ __forceinline__ void trace_tail(Ray& q, vec3 h, const PlaneK& P) {
    q.phat = P.phat;                                              // :60 (to_k: same roundings, once per launch)
    q.denom = dot3(q.r, q.phat);                                  // :62
    q.valid = fabsf(q.denom) > 1e-9f;                             // :63
    float safe = q.valid ? q.denom : 1e-9f;                       // :65
    q.t = __fdiv_rn(dot3(sub3(P.o, h), q.phat), safe);            // :67
    float st = q.valid ? q.t : 0.0f;                              // :69
    q.x = add3(h, scale3(st, q.r));                               // :71
    if (!q.valid) q.x = {0.0f, 0.0f, 0.0f};                       // :73
    // per-ray constants of the footprint, :126-127 and :146
    q.dh = sub3(q.x, h);
    q.dist = norm3(q.dh);
    q.sraw = __fmul_rn(P.sigma_scale, q.dist);
    q.sigma = clamp_min_t(q.sraw, 1e-9f);
    q.two_raw = __fmul_rn(2.0f, __fmul_rn(q.sigma, q.sigma));
    q.two_s2 = clamp_min_t(q.two_raw, 1e-12f);
    // separable restatement of :134-148: with d0 = o - x and the
    // orthonormal frame (u, v, w = u×v),  |P_ij - x|² = (xs_i+a)² + (ys_j+b)² + c²
    q.d0 = sub3(P.o, q.x);
    q.a = dot3(q.d0, P.u);
    q.b = dot3(q.d0, P.v);
    q.c = dot3(q.d0, P.w);
    q.c2 = __fmul_rn(q.c, q.c);
    q.k2 = q.valid ? __fdiv_rn(1.44269504088896340736f, q.two_s2) : 0.0f;
}

__device__ __forceinline__ Ray trace(vec3 nin, float ce, float se, float cu, float su,
                                     vec3 h, vec3 s, const PlaneK& P) {
    Ray q;
    trace_head(q, nin, ce, se, cu, su, h, s);
    trace_tail(q, h, P);
    return q;
}

static inline PlaneK to_k(const helio_plane* p) {
    PlaneK k;
    k.o = {p->origin[0], p->origin[1], p->origin[2]};
    k.nrm = {p->normal[0], p->normal[1], p->normal[2]};
    {   // unit3() on the host: IEEE fmaf / sqrtf / division round exactly as the device's (-ffp-contract=off)
        const float nn = sqrtf(fmaf(k.nrm.z, k.nrm.z, fmaf(k.nrm.y, k.nrm.y, k.nrm.x * k.nrm.x)));
        const float n = nn != nn ? nn : fmaxf(nn, 1e-9f);
        k.phat = {k.nrm.x / n, k.nrm.y / n, k.nrm.z / n};
    }
    k.u = {p->u[0], p->u[1], p->u[2]};
    k.v = {p->v[0], p->v[1], p->v[2]};
    k.w = {p->w[0], p->w[1], p->w[2]};
    k.sigma_scale = p->sigma_scale;
    return k;
}

}  // namespace helio
