// A torch-free client of the C ABI (include/helio.h): plain HIP runtime calls, raw device
// pointers, checked against the C oracle (oracle/helio_oracle.c) in the same program.
// Built and run by tests/test_c_abi_gpu.py:
//   gcc -O2 -ffp-contract=off -c oracle/helio_oracle.c -o oracle.o
//   g++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/c/abi_smoke.cpp oracle.o -I include \
//       -L doodle_amd -lhelio -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/doodle_amd
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "helio.h"

extern "C" {
void oracle_geometry(int B, int N, const float* helios, const float* sun, const float* action, const float* trig,
                     const float* target_pos, const float* target_normal, float* actual, float* refl, float* inter,
                     float* mask);
void oracle_splat(int B, int N, int R, const float* inter, const float* mask, const float* helios, const float* origin,
                  const float* u, const float* v, const float* xs, const float* ys, float sigma_scale, float* image);
}

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define ABI_OK(x) do { int rc_ = (x); if (rc_ != 0) { printf("ABI error %d: %s at %d\n", rc_, helio_last_error_string(), __LINE__); return 3; } } while (0)

static unsigned long long lcg = 88172645463325252ull;
static float rnd() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (float)((lcg >> 40) & 0xFFFFFF) / 16777216.0f; }

template <class T> static T* to_device(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 6, N = argc > 2 ? atoi(argv[2]) : 37, R = argc > 3 ? atoi(argv[3]) : 96;
    const float sigma = 0.03f;
    std::vector<float> helios(3 * N), sun(3 * B), action(3 * B * N), trig(4 * B * N), xs(R), ys(R);
    for (int n = 0; n < N; ++n) { helios[3 * n] = 80 + 10 * rnd(); helios[3 * n + 1] = 80 + 10 * rnd(); helios[3 * n + 2] = 0; }
    for (int b = 0; b < B; ++b) {
        float d[3] = {rnd() - 0.5f, rnd() - 0.5f, 0.3f + rnd()};
        float nrm = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int k = 0; k < 3; ++k) sun[3 * b + k] = d[k] / nrm * 14142.1356f;
    }
    const float tp[3] = {0.f, -5.f, 0.f}, tn[3] = {0.f, 1.f, 0.f}, u[3] = {1.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 1.f};
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {            // roughly the ideal normal, perturbed
            float a[3], c[3], s[3];
            float na = 0, nc = 0;
            for (int k = 0; k < 3; ++k) { a[k] = sun[3 * b + k] - helios[3 * n + k]; c[k] = tp[k] - helios[3 * n + k]; na += a[k] * a[k]; nc += c[k] * c[k]; }
            float ns = 0;
            for (int k = 0; k < 3; ++k) { s[k] = a[k] / std::sqrt(na) + c[k] / std::sqrt(nc) + 0.02f * (rnd() - 0.5f); ns += s[k] * s[k]; }
            for (int k = 0; k < 3; ++k) action[3 * (b * N + n) + k] = s[k] / std::sqrt(ns);
            const float e0 = (rnd() - 0.5f) * 0.08f, e1 = (rnd() - 0.5f) * 0.08f;     // radians
            float* t = &trig[4 * (b * N + n)];
            t[0] = std::cos(e0); t[1] = std::sin(e0); t[2] = std::cos(e1); t[3] = std::sin(e1);
        }
    for (int i = 0; i < R; ++i) xs[i] = ys[i] = -7.5f + 15.0f * (float)i / (float)(R > 1 ? R - 1 : 1);

    // oracle (CPU, scalar C)
    std::vector<float> o_actual(3 * B * N), o_refl(3 * B * N), o_inter(3 * B * N), o_mask(B * N), o_img((size_t)B * R * R);
    oracle_geometry(B, N, helios.data(), sun.data(), action.data(), trig.data(), tp, tn, o_actual.data(), o_refl.data(),
                    o_inter.data(), o_mask.data());
    oracle_splat(B, N, R, o_inter.data(), o_mask.data(), helios.data(), tp, u, v, xs.data(), ys.data(), sigma, o_img.data());

    // product (MI355X, through the C ABI)
    char arch[64] = "";
    ABI_OK(helio_device_arch(0, arch, 64));
    helio_plane plane;
    memcpy(plane.origin, tp, 12); memcpy(plane.normal, tn, 12); memcpy(plane.u, u, 12); memcpy(plane.v, v, 12);
    plane.w[0] = 0.f; plane.w[1] = -1.f; plane.w[2] = 0.f;      // u × v
    plane.sigma_scale = sigma;
    float *d_h = to_device(helios), *d_s = to_device(sun), *d_a = to_device(action), *d_t = to_device(trig),
          *d_x = to_device(xs), *d_y = to_device(ys), *d_actual, *d_refl, *d_rays, *d_img;
    HIP_OK(hipMalloc(&d_actual, 12ul * B * N)); HIP_OK(hipMalloc(&d_refl, 12ul * B * N));
    HIP_OK(hipMalloc(&d_rays, 16ul * B * N)); HIP_OK(hipMalloc(&d_img, 4ul * B * R * R));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    std::vector<float> actual(3 * B * N), refl(3 * B * N), img((size_t)B * R * R);
    int worst = 0;
    const int variants[] = {0, 1, 3, 4, 5, 6};
    for (int variant : variants) {
        // the optional device scratch (helio.h "Device scratch"): with it the large-problem kernels skip the rays
        // that are exactly zero on a tile; 0 bytes where the kernel this call runs takes none
        const long sbytes = helio_fwd_scratch_bytes(B, N, R, variant);
        void* d_scratch = nullptr;
        if (sbytes > 0) HIP_OK(hipMalloc(&d_scratch, (size_t)sbytes));
        ABI_OK(helio_render_fwd(B, N, R, d_h, d_s, d_a, d_t, 4l * N, &plane, d_x, d_y, d_actual, d_refl, d_rays, d_img,
                                variant, d_scratch, sbytes, st));
        HIP_OK(hipStreamSynchronize(st));
        if (d_scratch) {            // and the same call without scratch gives the same bits
            std::vector<float> with(img.size()), without(img.size());
            HIP_OK(hipMemcpy(with.data(), d_img, 4ul * B * R * R, hipMemcpyDeviceToHost));
            ABI_OK(helio_render_fwd(B, N, R, d_h, d_s, d_a, d_t, 4l * N, &plane, d_x, d_y, d_actual, d_refl, d_rays, d_img,
                                    variant, nullptr, 0, st));
            HIP_OK(hipStreamSynchronize(st));
            HIP_OK(hipMemcpy(without.data(), d_img, 4ul * B * R * R, hipMemcpyDeviceToHost));
            const bool same = memcmp(with.data(), without.data(), 4ul * B * R * R) == 0;
            printf("variant %d: %ld bytes of scratch, image %s the dense kernel's\n", variant, sbytes, same ? "bit-identical with" : "DIFFERS from");
            worst |= !same;
            HIP_OK(hipFree(d_scratch));
        }
        HIP_OK(hipMemcpy(actual.data(), d_actual, 12ul * B * N, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(refl.data(), d_refl, 12ul * B * N, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(img.data(), d_img, 4ul * B * R * R, hipMemcpyDeviceToHost));
        const bool bits = memcmp(actual.data(), o_actual.data(), 12ul * B * N) == 0 &&
                          memcmp(refl.data(), o_refl.data(), 12ul * B * N) == 0;
        double peak = 0, maxd = 0;
        long bad = 0;
        for (size_t p = 0; p < img.size(); ++p) {
            peak = std::fmax(peak, std::fabs(o_img[p]));
            const double d = std::fabs((double)img[p] - o_img[p]);
            maxd = std::fmax(maxd, d);
            if (d > 1e-8 + 1e-5 * std::fabs(o_img[p])) ++bad;
        }
        const bool ok = bits && bad == 0 && maxd <= 1e-5 * peak;
        printf("variant %d on %s: geometry %s, image max|d|/peak %.2e, %ld pixels out of tolerance -> %s\n", variant,
               arch, bits ? "bit-exact" : "DIFFERS", maxd / peak, bad, ok ? "ok" : "FAIL");
        worst |= !ok;
    }
    // HelioEnv.step forward in one call (helio_env_step_fwd) with a completion record: the image
    // it writes equals helio_render_fwd's bit for bit; mse / dist / per-image mean error agree with
    // a scalar restatement of test_environment.py:436-457 on the oracle image; the NaN/Inf flag
    // arrives through pinned host memory (helio_notify_wait), no device read.
    {
        std::vector<float> target((size_t)B * R * R), dmap((size_t)B * R * R), tx(B), ideal(3 * B * N);
        for (size_t p = 0; p < target.size(); ++p) { target[p] = 2.0f * rnd(); dmap[p] = 30.0f * rnd(); }
        for (int b = 0; b < B; ++b) {
            float m = 1e-6f;
            for (size_t p = 0; p < (size_t)R * R; ++p) m = std::fmax(m, target[(size_t)b * R * R + p]);
            tx[b] = m;
        }
        for (size_t k = 0; k < ideal.size(); ++k) ideal[k] = o_actual[k];       // any unit vectors will do
        float *d_target = to_device(target), *d_dm = to_device(dmap), *d_tx = to_device(tx), *d_ideal = to_device(ideal);
        float *d_ws, *d_out, *d_mae, *d_keep, *d_align, *d_allb, *d_img2;
        HIP_OK(hipMalloc(&d_ws, 4ul * helio_env_step_workspace(B, N, R))); HIP_OK(hipMalloc(&d_out, 20));
        HIP_OK(hipMalloc(&d_mae, 4ul * B)); HIP_OK(hipMalloc(&d_keep, 4ul * B)); HIP_OK(hipMalloc(&d_align, 4ul * B * N));
        HIP_OK(hipMalloc(&d_allb, 4ul * B * N)); HIP_OK(hipMalloc(&d_img2, 4ul * B * R * R));
        int* record = nullptr;
        ABI_OK(helio_notify_create(&record));
        ABI_OK(helio_render_fwd(B, N, R, d_h, d_s, d_a, d_t, 4l * N, &plane, d_x, d_y, d_actual, d_refl, d_rays, d_img, 0, nullptr, 0, st));
        ABI_OK(helio_env_step_fwd(B, N, R, d_h, d_s, d_a, d_t, 4l * N, &plane, d_x, d_y, d_actual, d_refl, d_rays, d_img2, 0,
                                  d_target, d_tx, d_dm, d_ideal, tp, tn, 15.0f, 15.0f, 0, -1.0f, d_ws, d_out, d_mae, d_keep,
                                  d_align, d_allb, nullptr, record, 7, nullptr, 0, st));
        const int flag = helio_notify_wait(record, 7, 10.0);       // returns once the finishing workgroup has published
        std::vector<float> img2((size_t)B * R * R), mae(B);
        float out[5];
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpy(img.data(), d_img, 4ul * B * R * R, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(img2.data(), d_img2, 4ul * B * R * R, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(out, d_out, 20, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(mae.data(), d_mae, 4ul * B, hipMemcpyDeviceToHost));
        double sq = 0, ds = 0, worst_mae = 0;
        for (int b = 0; b < B; ++b) {
            double ab = 0, dsb = 0;
            for (size_t p = 0; p < (size_t)R * R; ++p) {
                const size_t q = (size_t)b * R * R + p;
                const float d = o_img[q] / tx[b] - target[q] / tx[b];
                sq += (double)d * d; ab += std::fabs(d); dsb += std::fabs(d) * (double)dmap[q];
            }
            ds += dsb;
            worst_mae = std::fmax(worst_mae, std::fabs(mae[b] - ab / ((double)R * R)) / (ab / ((double)R * R)));
        }
        const double mse = sq / ((double)B * R * R), dist = ds / B;
        const bool same = memcmp(img.data(), img2.data(), 4ul * B * R * R) == 0;
        const bool ok = same && flag == 0 && out[4] == 0.0f && std::fabs(out[0] - mse) <= 2e-5 * mse &&
                        std::fabs(out[1] - dist) <= 2e-5 * dist && worst_mae <= 2e-5 &&
                        helio_env_step_launches(B, N, R) == (helio_render_fwd_launches(B, N, R) == 1 ? 2 : 4) &&
                        helio_notify_wait(record, 8, 0.01) == HELIO_E_TIMEOUT;
        printf("env step (%d launches): image %s, mse %.6e vs %.6e, dist %.6e vs %.6e, mae rel err %.1e, flag %d -> %s\n",
               helio_env_step_launches(B, N, R), same ? "bit-identical" : "DIFFERS", out[0], mse, out[1], dist, worst_mae,
               flag, ok ? "ok" : "FAIL");
        worst |= !ok;
        ABI_OK(helio_notify_destroy(record));
    }
    // error behaviour: invalid arguments are refused before any launch
    if (helio_render_fwd(0, N, R, d_h, d_s, d_a, d_t, 4l * N, &plane, d_x, d_y, d_actual, d_refl, d_rays, d_img, 0, nullptr, 0, st) != HELIO_E_INVALID) worst = 1;
    if (helio_render_fwd(B, N, R, nullptr, d_s, d_a, d_t, 4l * N, &plane, d_x, d_y, d_actual, d_refl, d_rays, d_img, 0, nullptr, 0, st) != HELIO_E_INVALID) worst = 1;
    printf(worst ? "ABI SMOKE FAILED\n" : "ABI SMOKE OK\n");
    return worst;
}
