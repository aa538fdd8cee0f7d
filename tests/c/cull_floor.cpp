// CPU check of the culling criterion's arithmetic (doodle_amd/csrc/cull_math.h), by brute force and bit for bit.
//
// The criterion drops a ray from a tile when exponent_floor_x + exponent_floor_y exceeds a threshold; it is only
// sound if exponent_floor() really is a LOWER BOUND of every exponent the footprint kernels compute for a pixel of
// the tile, in each of the three ways they form a factor (cull.h).  Every operation involved — fmaf, +, *, sqrtf — is
// a correctly rounded IEEE operation on the CPU and on gfx950 alike (-ffp-contract=off on both sides), so the
// claim can be checked here exactly: for random rays and coordinate arrays — ascending like torch.linspace,
// descending, and shuffled (the C ABI takes any array) — the exponent of EVERY pixel in EVERY form is compared with
// the floor.  Also: NaN anywhere keeps the ray, a plane-parallel ray (k2 = 0) is never dropped.
// Build: g++ -O1 -ffp-contract=off -std=c++17 -I doodle_amd/csrc tests/c/cull_floor.cpp   (tests/test_cull_math.py)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define HELIO_HD static inline
#include "cull_math.h"

using namespace helio;

int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    long checked = 0, dead_product = 0, dead_strict = 0, violations = 0;
    for (int trial = 0; trial < 4000; ++trial) {
        const int R = 1 + (int)(U(rng) * 300);
        const float W = (float)(0.5 + U(rng) * 40.0), H = (float)(0.5 + U(rng) * 40.0);
        std::vector<float> xs(R), ys(R);
        for (int i = 0; i < R; ++i) {
            xs[i] = R > 1 ? -W / 2 + W * (float)i / (float)(R - 1) : 0.0f;
            ys[i] = R > 1 ? -H / 2 + H * (float)i / (float)(R - 1) : 0.0f;
        }
        const int order = trial % 3;                               // ascending, descending, shuffled
        if (order == 1) { std::reverse(xs.begin(), xs.end()); std::reverse(ys.begin(), ys.end()); }
        if (order == 2) { std::shuffle(xs.begin(), xs.end(), rng); std::shuffle(ys.begin(), ys.end(), rng); }
        // a tile of the image: [i0, i1) x [j0, j1)
        const int i0 = (int)(U(rng) * R), i1 = std::min(R, i0 + 1 + (int)(U(rng) * 256));
        const int j0 = (int)(U(rng) * R), j1 = std::min(R, j0 + 1 + (int)(U(rng) * 256));
        CullBox bx;
        bx.xlo = *std::min_element(xs.begin() + i0, xs.begin() + i1); bx.xhi = *std::max_element(xs.begin() + i0, xs.begin() + i1);
        bx.ylo = *std::min_element(ys.begin() + j0, ys.begin() + j1); bx.yhi = *std::max_element(ys.begin() + j0, ys.begin() + j1);
        for (int r = 0; r < 40; ++r) {
            // rays around the threshold, on the tile, far away, with huge / tiny sigma
            const float k2 = (float)std::pow(10.0, U(rng) * 16.0 - 6.0);                  // 1e-6 … 1e10 (the clamp allows 1.4e12)
            const double reach = std::sqrt((100.0 + 100.0 * U(rng)) / k2);               // distance whose exponent is 100 … 200
            const float a = (float)(-(bx.xlo + bx.xhi) / 2 + (U(rng) < 0.5 ? -1 : 1) * (U(rng) < 0.3 ? U(rng) * W : (W / 2 + reach * U(rng) * 1.5)));
            const float b = (float)(-(bx.ylo + bx.yhi) / 2 + (U(rng) < 0.5 ? -1 : 1) * (U(rng) < 0.3 ? U(rng) * H : (H / 2 + reach * U(rng) * 1.5)));
            const float c2 = U(rng) < 0.5 ? 0.0f : (float)(U(rng) * 30.0 / k2);
            float fx, fy;
            cull_floors(a, b, k2, c2, bx, fx, fy);
            const float sk = __builtin_sqrtf(k2);
            // every exponent the kernels compute on this tile, in the three forms (cull.h)
            float min_x = INFINITY, min_y = INFINITY;
            for (int i = i0; i < i1; ++i) {
                const float q = __builtin_fmaf(xs[i], sk, a * sk), t = xs[i] + a;
                min_x = std::min({min_x, __builtin_fmaf(q, q, c2 * k2), __builtin_fmaf(t, t, c2) * k2, ((t * t) + c2) * k2});
            }
            for (int j = j0; j < j1; ++j) {
                const float q = __builtin_fmaf(ys[j], sk, b * sk), t = ys[j] + b;
                min_y = std::min({min_y, __builtin_fmaf(q, q, 0.0f * k2), __builtin_fmaf(t, t, 0.0f) * k2, ((t * t) + 0.0f) * k2});
            }
            ++checked;
            if (!(fx <= min_x) || !(fy <= min_y)) {
                if (violations++ < 10)
                    printf("VIOLATION: floor (%g, %g) above a computed exponent (%g, %g): a=%g b=%g k2=%g c2=%g box x[%g,%g] y[%g,%g]\n",
                           fx, fy, min_x, min_y, a, b, k2, c2, bx.xlo, bx.xhi, bx.ylo, bx.yhi);
            }
            const bool dp = cull_dead_product(a, b, k2, c2, bx), ds = cull_dead_strict(a, b, k2, c2, bx);
            dead_product += dp; dead_strict += ds;
            if (ds && !dp && fx + fy == fx + fy) { printf("VIOLATION: strictly dead but not product-dead\n"); ++violations; }
            if (dp && !(min_x + min_y > CULL_EXP2 - 1e-3f)) { printf("VIOLATION: dropped with a pixel exponent of %g\n", min_x + min_y); ++violations; }
        }
        // never dropped: plane-parallel rays (k2 = 0), NaN parameters, NaN coordinates
        const float nan = NAN;
        const bool bad = cull_dead_product(3.f, 4.f, 0.0f, 2.f, bx) || cull_dead_strict(1e30f, 4.f, 0.0f, 2.f, bx) ||
                         cull_dead_product(nan, 0.f, 1.f, 0.f, bx) || cull_dead_strict(0.f, nan, 1.f, 0.f, bx) ||
                         cull_dead_product(1e9f, 1e9f, nan, 0.f, bx) || cull_dead_strict(1e9f, 1e9f, 1.f, nan, bx);
        if (bad) { printf("VIOLATION: a ray that must be kept was dropped\n"); ++violations; }
        CullBox nb = bx; nb.xlo = nb.xhi = nan;                 // what block_minmax reports for a tile with a NaN coordinate
        CullBox nc = bx; nc.yhi = nan;
        if (cull_dead_product(1e9f, 1e9f, 1.f, 0.f, nb) || cull_dead_strict(1e9f, 1e9f, 1.f, 0.f, nb) ||
            cull_dead_product(1e9f, 1e9f, 1.f, 0.f, nc) || cull_dead_strict(1e9f, 1e9f, 1.f, 0.f, nc)) {
            printf("VIOLATION: dropped on a NaN coordinate box\n"); ++violations;
        }
    }
    printf("checked %ld (ray, tile) pairs: %ld dropped by the product criterion, %ld by the strict one, %ld violations\n",
           checked, dead_product, dead_strict, violations);
    printf(violations == 0 && dead_product > checked / 20 && dead_product < checked ? "CULL FLOOR OK\n" : "CULL FLOOR FAILED\n");
    return violations == 0 ? 0 : 1;
}
