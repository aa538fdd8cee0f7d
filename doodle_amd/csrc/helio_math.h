// Device-side arithmetic helpers shared by the geometry kernels.
//
// The geometry stage must be BIT-IDENTICAL with the reference's fp32 CPU results
// (an algebraically equal geometry with different rounding
// already breaks the 1e-5 image tolerance at sigma_scale=0.01).  So every
// primitive below restates exactly what the ATen CPU kernel does, and the
// library is compiled with -ffp-contract=off so that nothing else fuses:
//   norm over 3 elements  = sqrt(fma(z,z,fma(y,y,x*x)))      (correctly rounded)
//   dot over 3 elements   = (p0+p1)+p2, products rounded separately
//   x / max(norm,1e-9)    = IEEE division per component
// NB: HIP's __fsqrt_rn() is the APPROXIMATE v_sqrt_f32 (see __clang_hip_math.h); the
// correctly rounded square root is plain sqrtf under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt, which build.py passes explicitly.
#pragma once
#include <hip/hip_runtime.h>

namespace helio {

struct vec3 { float x, y, z; };

__device__ __forceinline__ vec3 ld3(const float* __restrict__ p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* __restrict__ p, vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

__device__ __forceinline__ float norm3(vec3 v) {
    return __builtin_sqrtf(__builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
}
__device__ __forceinline__ float dot3(vec3 a, vec3 b) {
    return __fadd_rn(__fadd_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)), __fmul_rn(a.z, b.z));
}
__device__ __forceinline__ vec3 sub3(vec3 a, vec3 b) { return {__fsub_rn(a.x, b.x), __fsub_rn(a.y, b.y), __fsub_rn(a.z, b.z)}; }
__device__ __forceinline__ vec3 add3(vec3 a, vec3 b) { return {__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y), __fadd_rn(a.z, b.z)}; }
__device__ __forceinline__ vec3 scale3(float s, vec3 a) { return {__fmul_rn(s, a.x), __fmul_rn(s, a.y), __fmul_rn(s, a.z)}; }
// torch's clamp_min keeps a NaN (fmaxf would replace it by the bound)
__device__ __forceinline__ float clamp_min_t(float x, float lo) { return x != x ? x : fmaxf(x, lo); }

// x / max(|x|, 1e-9); also hands back the clamped norm
__device__ __forceinline__ vec3 unit3(vec3 v, float& nclamped) {
    float n = clamp_min_t(norm3(v), 1e-9f);
    nclamped = n;
    return {__fdiv_rn(v.x, n), __fdiv_rn(v.y, n), __fdiv_rn(v.z, n)};
}
__device__ __forceinline__ vec3 unit3(vec3 v) { float n; return unit3(v, n); }

}  // namespace helio
