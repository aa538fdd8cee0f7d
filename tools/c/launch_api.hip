// Host cost of one kernel launch through the HIP entry points a library can use, for a kernel with the
// argument block of render_fwd_fused_small (≈200 bytes).  hipcc --offload-arch=gfx950 -O2 launch_api.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>

struct Late { float p[21]; float *a, *b, *c, *d; };
__global__ void k_args(int N, int R, const float* h, const float* s, const float* a, const float* t, const float* x,
                       const float* y, Late late, float* out) {
    if (N < 0) out[0] = late.p[0] + h[0] + s[0] + a[0] + t[0] + x[0] + y[0] + R;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000; const int g = argc > 2 ? atoi(argv[2]) : 4;
    float* buf; CK(hipMalloc(&buf, 1024));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Late late; memset(&late, 0, sizeof late);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ns = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::nano>(now() - t0).count() / n; };
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_args, dim3(g, g, 25), dim3(256), 0, st, 1, 2, buf, buf, buf, buf, buf, buf, late, buf);
        CK(hipStreamSynchronize(st));
        auto t0 = now();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_args, dim3(g, g, 25), dim3(256), 0, st, 1, 2, buf, buf, buf, buf, buf, buf, late, buf);
        double a = ns(t0); CK(hipStreamSynchronize(st)); double a2 = ns(t0);

        hipFunction_t fn; CK(hipGetFuncBySymbol(&fn, (const void*)k_args));
        int N = 1, R = 2; const float* p = buf; float* o = buf;
        void* params[] = {&N, &R, &p, &p, &p, &p, &p, &p, &late, &o};
        t0 = now();
        for (int i = 0; i < n; ++i) CK(hipModuleLaunchKernel(fn, g, g, 25, 256, 1, 1, 0, st, params, nullptr));
        double b = ns(t0); CK(hipStreamSynchronize(st)); double b2 = ns(t0);

        struct __attribute__((packed, aligned(8))) Packed { int N, R; const float* p[6]; Late late; float* o; } pk;
        pk.N = 1; pk.R = 2; for (auto& q : pk.p) q = buf; pk.late = late; pk.o = buf;
        size_t sz = sizeof pk;
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        t0 = now();
        for (int i = 0; i < n; ++i) CK(hipModuleLaunchKernel(fn, g, g, 25, 256, 1, 1, 0, st, nullptr, extra));
        double c = ns(t0); CK(hipStreamSynchronize(st)); double c2 = ns(t0);

        t0 = now();
        for (int i = 0; i < n; ++i) CK(hipLaunchKernel((const void*)k_args, dim3(g, g, 25), dim3(256), params, 0, st));
        double d = ns(t0); CK(hipStreamSynchronize(st)); double d2 = ns(t0);
        printf("ns per launch (enqueue / with drain): hipLaunchKernelGGL %.0f / %.0f | hipModuleLaunchKernel(params) %.0f / %.0f | "
               "hipModuleLaunchKernel(extra buffer) %.0f / %.0f | hipLaunchKernel %.0f / %.0f\n", a, a2, b, b2, c, c2, d, d2);
    }
    return 0;
}
