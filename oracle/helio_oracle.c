/*
 * helio_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar, single-threaded C restatement of the reference's heliostat render
 * (DOODLE newenv_rl_test_multi_error.py).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's library; doodle_amd/
 * never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * the fixtures in tests/golden/, which were produced by running the reference
 * itself (tests/golden/make_golden.py):  `actual`, `refl`, the intersection
 * points and the validity mask match BIT FOR BIT; images match within
 * rtol 1e-5 / atol 1e-8 (glibc expf vs torch's SLEEF expf, and fp64 vs torch's
 * cascade summation over heliostats, are the only differences).
 *
 * Arithmetic rules that make the geometry bit-exact with torch's CPU kernels
 * (measured in SURVEY.md Appendix A; build with -ffp-contract=off):
 *   - norm over the last dim of an [M,3] tensor = sqrtf(fmaf(z,z,fmaf(y,y,x*x)))
 *   - (a*b).sum(dim=1) over 3 elements = (p0+p1)+p2, products rounded separately
 *   - every other mul/add/sub/div is individually rounded, IEEE division
 *   - cos/sin of the error angles and the linspace pixel coordinates are INPUTS
 *     (taken from host torch; torch's CPU linspace/trig are SIMD kernels)
 *
 * Each function cites the reference lines it follows (file = reference's
 * newenv_rl_test_multi_error.py unless stated).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

static inline float norm3(const float v[3]) {
    /* torch .norm(dim=1) on [M,3]: an FMA chain, see header */
    return sqrtf(fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
}

static inline float dot3(const float a[3], const float b[3]) {
    /* (a*b).sum(dim=1): products rounded, summed left to right */
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

static inline void unit3(const float v[3], float out[3]) {
    /* x / x.norm(...).clamp_min(1e-9) */
    float n = norm3(v);
    if (n < 1e-9f) n = 1e-9f;
    out[0] = v[0] / n; out[1] = v[1] / n; out[2] = v[2] / n;
}

/*
 * Ray geometry for every (sun b, heliostat n):  :78-104 (error rotation),
 * :369-373 (leaky-ReLU on Z + renormalise), :376-380 (incident direction),
 * :46-50 + :383 (reflection, renormalised), :52-75 (ray/plane intersection).
 *
 *   helios[N,3]  sun[B,3]  action[B,N,3]
 *   trig[B,N,4] = (cos_e, sin_e, cos_u, sin_u) of errs*1e-3  (host torch)
 *   target_pos[3], target_normal[3] (as stored by the field ctor, :192)
 * outputs (any may be NULL):
 *   actual[B,N,3]  refl[B,N,3]  inter[B,N,3]  mask[B,N]
 */
void oracle_geometry(int B, int N,
                     const float *helios, const float *sun, const float *action,
                     const float *trig,
                     const float *target_pos, const float *target_normal,
                     float *actual, float *refl, float *inter, float *mask)
{
    float phat[3];
    unit3(target_normal, phat);                    /* :60 (divided again)       */
    for (int b = 0; b < B; ++b) {
        for (int n = 0; n < N; ++n) {
            size_t m = (size_t)b * N + n;
            const float *a = action + 3 * m, *t = trig + 4 * m, *h = helios + 3 * n;
            float ce = t[0], se = t[1], cu = t[2], su = t[3];
            float x = a[0], y = a[1], z = a[2];
            /* :96-102  rotate about Up (Z), then about East (X) */
            float xu = cu * x - su * y;
            float yu = su * x + cu * y;
            float ye = ce * yu - se * z;
            float ze = se * yu + ce * z;
            /* :369  F.leaky_relu, default slope 0.01 */
            ze = ze > 0.0f ? ze : ze * 0.01f;
            float v[3] = {xu, ye, ze}, act[3];
            unit3(v, act);                         /* :372 */
            /* :377-380 incident direction */
            float d[3] = {sun[3 * b] - h[0], sun[3 * b + 1] - h[1], sun[3 * b + 2] - h[2]};
            float inc[3];
            unit3(d, inc);
            /* :48-50 reflect about the (re-normalised) normal */
            float nh[3];
            unit3(act, nh);
            float dots = -dot3(inc, nh);
            float two = 2.0f * dots;
            float r0[3] = {(-inc[0]) - two * nh[0], (-inc[1]) - two * nh[1],
                           (-inc[2]) - two * nh[2]};
            float r[3];
            unit3(r0, r);                          /* :383 */
            /* :62-73 intersection with the target plane */
            float denom = dot3(r, phat);
            int valid = fabsf(denom) > 1e-9f;
            float safe = valid ? denom : 1e-9f;
            float ph[3] = {target_pos[0] - h[0], target_pos[1] - h[1], target_pos[2] - h[2]};
            float tt = dot3(ph, phat) / safe;
            float st = valid ? tt : 0.0f;
            float xi[3] = {h[0] + st * r[0], h[1] + st * r[1], h[2] + st * r[2]};
            if (!valid) xi[0] = xi[1] = xi[2] = 0.0f;
            if (actual) { actual[3 * m] = act[0]; actual[3 * m + 1] = act[1]; actual[3 * m + 2] = act[2]; }
            if (refl)   { refl[3 * m] = r[0];     refl[3 * m + 1] = r[1];     refl[3 * m + 2] = r[2]; }
            if (inter)  { inter[3 * m] = xi[0];   inter[3 * m + 1] = xi[1];   inter[3 * m + 2] = xi[2]; }
            if (mask)   mask[m] = valid ? 1.0f : 0.0f;
        }
    }
}

/*
 * Gaussian footprints summed over heliostats:  :107-149 and :404-406.
 *   image[b,i,j] = sum_n exp( -|((o + xs[i]*u) + ys[j]*v - x_bn) * mask_bn|^2
 *                             / max(2*sigma_bn^2, 1e-12) ),
 *   sigma_bn = max(sigma_scale * |x_bn - h_n|, 1e-9).
 * Image dim0 (i) runs along plane_u, dim1 (j) along plane_v.  The sum over n is
 * carried in double and rounded once (torch uses a float cascade sum).
 */
void oracle_splat(int B, int N, int R,
                  const float *inter, const float *mask, const float *helios,
                  const float *origin, const float *u, const float *v,
                  const float *xs, const float *ys, float sigma_scale,
                  float *image)
{
    double *acc = (double *)malloc(sizeof(double) * (size_t)R * R);
    for (int b = 0; b < B; ++b) {
        for (size_t p = 0; p < (size_t)R * R; ++p) acc[p] = 0.0;
        for (int n = 0; n < N; ++n) {
            size_t m = (size_t)b * N + n;
            const float *x = inter + 3 * m, *h = helios + 3 * n;
            float dh[3] = {x[0] - h[0], x[1] - h[1], x[2] - h[2]};
            float sigma = sigma_scale * norm3(dh);             /* :126-127 */
            if (sigma < 1e-9f) sigma = 1e-9f;
            float two_s2 = 2.0f * (sigma * sigma);             /* :146 */
            if (two_s2 < 1e-12f) two_s2 = 1e-12f;
            float mk = mask[m];
            for (int i = 0; i < R; ++i) {
                for (int j = 0; j < R; ++j) {
                    float D = 0.0f;
                    float dd[3];
                    for (int c = 0; c < 3; ++c) {
                        float P = (origin[c] + xs[i] * u[c]) + ys[j] * v[c];   /* :134-138 */
                        dd[c] = (P - x[c]) * mk;                              /* :142-143 */
                    }
                    D = (dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2];      /* :145 */
                    acc[(size_t)i * R + j] += (double)expf(-D / two_s2);     /* :148 */
                }
            }
        }
        for (size_t p = 0; p < (size_t)R * R; ++p)
            image[(size_t)b * R * R + p] = (float)acc[p];
    }
    free(acc);
}

/*
 * calculate_ideal_normals, :256-278:
 *   normalize( normalize(sun_b - h_n) + normalize(target - h_n) )
 */
void oracle_ideal_normals(int B, int N, const float *helios, const float *sun,
                          const float *target_pos, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const float *h = helios + 3 * n;
            float di[3] = {sun[3 * b] - h[0], sun[3 * b + 1] - h[1], sun[3 * b + 2] - h[2]};
            float dr[3] = {target_pos[0] - h[0], target_pos[1] - h[1], target_pos[2] - h[2]};
            float a[3], c[3];
            unit3(di, a);
            unit3(dr, c);
            float s[3] = {a[0] + c[0], a[1] + c[1], a[2] + c[2]};
            unit3(s, out + 3 * ((size_t)b * N + n));
        }
}

int oracle_version(void) { return 1; }
