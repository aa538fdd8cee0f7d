// The arithmetic of the culling criterion (cull.h), in a header of its own with NO device-specific include, so that
// the claim it rests on — exponent_floor() is a lower bound of every exponent the kernels compute over a tile — is
// also checked on the CPU, bit for bit, by brute force (tests/c/cull_floor.cpp, tests/test_cull_math.py: every
// operation here is a correctly rounded IEEE operation on both sides).  The includer defines HELIO_HD, the function
// qualifier: `__device__ __forceinline__` in the library, `static inline` in the host test.
#pragma once
#include <cmath>

namespace helio {

constexpr float CULL_EXP2 = 152.0f;

HELIO_HD float cull_nanmin(float a, float b) {
    return (a != a || b != b) ? __builtin_nanf("") : fminf(a, b);
}

// the point of [lo, hi] nearest to zero, as a magnitude (NaN if either end is: the comparisons alone would answer 0)
HELIO_HD float cull_nearest(float lo, float hi) {
    if (lo != lo || hi != hi) return __builtin_nanf("");
    return lo > 0.0f ? lo : (hi < 0.0f ? -hi : 0.0f);
}

// smallest exponent (base 2) any of the kernels' factor forms computes for a coordinate in [lo, hi]
HELIO_HD float exponent_floor(float lo, float hi, float shift, float k2, float sk, float cc) {
    if (lo != lo || hi != hi) return __builtin_nanf("");      // a NaN coordinate on the tile: no statement, every ray is kept
    const float ssk = shift * sk;
    const float qm = cull_nearest(__builtin_fmaf(lo, sk, ssk), __builtin_fmaf(hi, sk, ssk));
    const float eq = __builtin_fmaf(qm, qm, cc * k2);
    const float tm = cull_nearest(lo + shift, hi + shift);
    const float ef = __builtin_fmaf(tm, tm, cc) * k2;
    const float eu = ((tm * tm) + cc) * k2;
    return cull_nanmin(eq, cull_nanmin(ef, eu));
}

struct CullBox { float xlo, xhi, ylo, yhi; };

// ray = (a, b, k2, c2)
HELIO_HD void cull_floors(float a, float b, float k2, float c2, const CullBox& bx, float& fx, float& fy) {
    const float sk = __builtin_sqrtf(k2);
    fx = exponent_floor(bx.xlo, bx.xhi, a, k2, sk, c2);
    fy = exponent_floor(bx.ylo, bx.yhi, b, k2, sk, 0.0f);
}
HELIO_HD bool cull_dead_product(float a, float b, float k2, float c2, const CullBox& bx) {
    float fx, fy;
    cull_floors(a, b, k2, c2, bx, fx, fy);
    return fx + fy > CULL_EXP2;                       // false for NaN: kept
}
HELIO_HD bool cull_dead_strict(float a, float b, float k2, float c2, const CullBox& bx) {
    float fx, fy;
    cull_floors(a, b, k2, c2, bx, fx, fy);
    // both floors must be numbers: a NaN on one axis makes the dense kernels' 0·NaN products NaN, which a dropped
    // ray would hide
    return fx == fx && fy == fy && (fx > CULL_EXP2 || fy > CULL_EXP2);
}

}  // namespace helio
