// Sanitizer run of the PRODUCT's host layer (CPU only, no device): doodle_amd/csrc/abi.hip — the
// extern "C" entry points of include/helio.h with their argument validation — is compiled as host
// C++ with -fsanitize=address,undefined and linked against launch stubs that tests/test_abi_sanitizers.py
// generates from abi.hip's own forward declarations (every helio::launch_* only counts the call).
// Contract checked here: a call with a null pointer, a bad size, a misaligned buffer, a bad stride or
// a reserved ticket returns HELIO_E_INVALID with a message and NEVER reaches a launch; well-formed
// calls do reach one.  No pointer is dereferenced by the host layer, so fake device addresses do.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "helio.h"

extern int g_launches;   // abi_san_stubs.cpp (generated)

static int failures = 0;

#define EXPECT_INVALID(call)                                                                         \
    do {                                                                                             \
        const int before = g_launches;                                                               \
        const int rc = (call);                                                                       \
        if (rc != HELIO_E_INVALID || g_launches != before || !helio_last_error_string()[0]) {        \
            printf("FAIL line %d: rc=%d launches %d->%d msg='%s'\n", __LINE__, rc, before, g_launches, \
                   helio_last_error_string());                                                       \
            ++failures;                                                                              \
        }                                                                                            \
    } while (0)

// a well-formed call: validation lets it through to (at least) one launch stub; without a device the
// hipGetLastError() behind it may or may not report, so both HELIO_OK and HELIO_E_LAUNCH are fine
#define EXPECT_LAUNCH(call)                                                                          \
    do {                                                                                             \
        const int before = g_launches;                                                               \
        const int rc = (call);                                                                       \
        if ((rc != HELIO_OK && rc != HELIO_E_LAUNCH) || g_launches == before) {                      \
            printf("FAIL line %d: rc=%d launches %d->%d msg='%s'\n", __LINE__, rc, before, g_launches, \
                   helio_last_error_string());                                                       \
            ++failures;                                                                              \
        }                                                                                            \
    } while (0)

int main() {
    // fake, 16-byte aligned "device" addresses; P1 is deliberately misaligned
    float* const P = reinterpret_cast<float*>(uintptr_t(0x7f0000001000));
    float* const P1 = reinterpret_cast<float*>(uintptr_t(0x7f0000001004));
    void* const WS = P;
    void* const SCR = reinterpret_cast<void*>(uintptr_t(0x7f0000004000));     // 256-byte aligned "device scratch"
    int* const REC = reinterpret_cast<int*>(uintptr_t(0x7f0000002000));
    helio_plane plane;
    memset(&plane, 0, sizeof plane);
    plane.normal[1] = plane.u[0] = plane.v[2] = 1.0f; plane.w[1] = -1.0f; plane.sigma_scale = 0.01f;
    const float t3[3] = {0.f, -5.f, 0.f}, n3[3] = {0.f, 1.f, 0.f};
    const int B = 3, N = 7, R = 33;
    const long S = 4l * N;

    if (helio_abi_version() != HELIO_ABI_VERSION) { printf("FAIL abi version\n"); ++failures; }
    char buf[8];
    if (helio_device_arch(9999, buf, sizeof buf) != HELIO_E_NODEVICE) { printf("FAIL device_arch\n"); ++failures; }

    // ---- helio_error_trig / helio_init_actions
    EXPECT_INVALID(helio_error_trig(0, P, P, nullptr));
    EXPECT_INVALID(helio_error_trig(1l << 40, P, P, nullptr));
    EXPECT_INVALID(helio_error_trig(5, nullptr, P, nullptr));
    EXPECT_INVALID(helio_error_trig(5, P, nullptr, nullptr));
    EXPECT_INVALID(helio_error_trig(5, P1, P, nullptr));
    EXPECT_INVALID(helio_error_trig(5, P, P1, nullptr));
    EXPECT_LAUNCH(helio_error_trig(5, P, P, nullptr));
    EXPECT_INVALID(helio_init_actions(0, P, P, 0.01f, P, nullptr));
    EXPECT_INVALID(helio_init_actions(5, nullptr, P, 0.01f, P, nullptr));
    EXPECT_INVALID(helio_init_actions(5, P, nullptr, 0.01f, P, nullptr));
    EXPECT_INVALID(helio_init_actions(5, P, P, 0.01f, nullptr, nullptr));
    EXPECT_LAUNCH(helio_init_actions(5, P, P, 0.01f, P, nullptr));

    // ---- helio_geometry_fwd
    EXPECT_INVALID(helio_geometry_fwd(0, N, P, P, P, P, S, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, 0, P, P, P, P, S, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(70000, N, P, P, P, P, S, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(60000, 60000, P, P, P, P, 4l * 60000, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, N, nullptr, P, P, P, S, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, N, P, nullptr, P, P, S, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, N, P, P, nullptr, P, S, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, N, P, P, P, nullptr, S, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, N, P, P, P, P, S, nullptr, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, N, P, P, P, P, S, &plane, nullptr, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, N, P, P, P, P, S + 1, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, N, P, P, P, P1, S, &plane, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_fwd(B, N, P, P, P, P, S, &plane, P, P, P1, nullptr));
    EXPECT_LAUNCH(helio_geometry_fwd(B, N, P, P, P, P, S, &plane, P, nullptr, nullptr, nullptr));
    EXPECT_LAUNCH(helio_geometry_fwd(B, N, P, P, P, P, 0, &plane, P, P, P, nullptr));

    // ---- helio_splat_fwd
    EXPECT_INVALID(helio_splat_fwd(B, N, 0, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_fwd(B, N, 20000, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_fwd(-1, N, R, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_fwd(B, N, R, nullptr, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_fwd(B, N, R, P, nullptr, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_fwd(B, N, R, P, P, nullptr, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_fwd(B, N, R, P, P, P, nullptr, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_fwd(B, N, R, P1, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_fwd(B, N, R, P, P, P, P1, 0, nullptr, 0, nullptr));
    EXPECT_LAUNCH(helio_splat_fwd(B, N, R, P, P, P, P, 0, nullptr, 0, nullptr));
    // the optional device scratch: NULL with any size is "none"; otherwise 256-byte aligned and a size >= 0; a
    // buffer that is too small is not an error (the dense kernels run)
    EXPECT_INVALID(helio_splat_fwd(B, N, R, P, P, P, P, 0, P1, 1 << 20, nullptr));
    EXPECT_INVALID(helio_splat_fwd(B, N, R, P, P, P, P, 0, reinterpret_cast<float*>(uintptr_t(0x7f0000001010)), 1 << 20, nullptr));
    EXPECT_INVALID(helio_splat_fwd(B, N, R, P, P, P, P, 0, SCR, -1, nullptr));
    EXPECT_LAUNCH(helio_splat_fwd(B, N, R, P, P, P, P, 0, nullptr, 12345, nullptr));
    EXPECT_LAUNCH(helio_splat_fwd(B, N, R, P, P, P, P, 0, SCR, 1 << 20, nullptr));
    EXPECT_LAUNCH(helio_splat_fwd(512, 2000, 512, P, P, P, P, 5, SCR, 16, nullptr));
    EXPECT_LAUNCH(helio_splat_fwd(512, 2000, 512, P, P, P, P, 5, SCR, helio_fwd_scratch_bytes(512, 2000, 512, 5), nullptr));
    if (helio_fwd_scratch_bytes(B, N, R, 0) != 0 || helio_fwd_scratch_bytes(0, N, R, 0) != 0 || helio_fwd_scratch_bytes(512, 2000, 512, 5) <= 0 ||
        helio_fwd_scratch_bytes(512, 2000, 512, 12) != 0 || helio_bwd_scratch_bytes(B, N, R, 0) != 0 ||
        helio_bwd_scratch_bytes(512, 2000, 0, 2) != 0 || helio_bwd_scratch_bytes(512, 2000, 512, 2) <= 0 ||
        helio_bwd_scratch_bytes(512, 2000, 512, 8) != 0) { printf("FAIL scratch sizes\n"); ++failures; }

    // ---- helio_render_fwd
    EXPECT_INVALID(helio_render_fwd(B, N, 0, P, P, P, P, S, &plane, P, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(0, N, R, P, P, P, P, S, &plane, P, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(B, N, R, nullptr, P, P, P, S, &plane, P, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(B, N, R, P, P, P, P, S, &plane, nullptr, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(B, N, R, P, P, P, P, S, &plane, P, nullptr, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(B, N, R, P, P, P, P, S, &plane, P, P, nullptr, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(B, N, R, P, P, P, P, S, &plane, P, P, P, P, P, nullptr, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(B, N, R, P, P, P, P, 3, &plane, P, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(B, N, R, P, P, P, P, S, &plane, P, P, P, P, P, P1, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(B, N, R, P, P, P, P, S, &plane, P, P, P, P, P1, P, 0, nullptr, 0, nullptr));
    EXPECT_LAUNCH(helio_render_fwd(B, N, R, P, P, P, P, S, &plane, P, P, P, nullptr, P, P, 0, nullptr, 0, nullptr));
    // a large problem takes geometry + splat and then NEEDS the ray work buffer
    EXPECT_INVALID(helio_render_fwd(512, 2000, 512, P, P, P, P, 8000, &plane, P, P, P, P, nullptr, P, 0, nullptr, 0, nullptr));
    EXPECT_LAUNCH(helio_render_fwd(512, 2000, 512, P, P, P, P, 8000, &plane, P, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_fwd(512, 2000, 512, P, P, P, P, 8000, &plane, P, P, P, P, P, P, 0, P1, 1 << 20, nullptr));
    EXPECT_INVALID(helio_render_fwd(512, 2000, 512, P, P, P, P, 8000, &plane, P, P, P, P, P, P, 0, SCR, -5, nullptr));
    EXPECT_LAUNCH(helio_render_fwd(512, 2000, 512, P, P, P, P, 8000, &plane, P, P, P, P, P, P, 0, SCR, 1l << 30, nullptr));
    // the split variants keep partial images in the scratch: without it, HELIO_E_SCRATCH and no launch
    {
        const int before = g_launches;
        const int rc = helio_render_fwd(32, 2000, 512, P, P, P, P, 8000, &plane, P, P, P, P, P, P, 16, nullptr, 0, nullptr);
        const int rc2 = helio_render_fwd(32, 2000, 512, P, P, P, P, 8000, &plane, P, P, P, P, P, P, 16, SCR, 100, nullptr);
        if (rc != HELIO_E_SCRATCH || rc2 != HELIO_E_SCRATCH || g_launches != before || !helio_last_error_string()[0]) {
            printf("FAIL split variant without scratch: rc=%d rc2=%d launches %d->%d\n", rc, rc2, before, g_launches);
            ++failures;
        }
        if (helio_fwd_scratch_required(32, 2000, 512, 16) <= 0 || helio_fwd_scratch_required(32, 2000, 512, 5) != 0) { printf("FAIL scratch required\n"); ++failures; }
    }
    EXPECT_LAUNCH(helio_render_fwd(32, 2000, 512, P, P, P, P, 8000, &plane, P, P, P, P, P, P, 16, SCR, 1l << 30, nullptr));
    if (helio_render_fwd_launches(B, N, R) != 1 || helio_render_fwd_launches(512, 2000, 512) != 2) { printf("FAIL launches\n"); ++failures; }
    if (helio_render_fwd_choice(B, N, R) != 12 || helio_render_fwd_choice(512, 2000, 512) != 6 || helio_render_fwd_choice(0, N, R) != 0 ||
        helio_render_fwd_choice(B, N, 0) != 0) { printf("FAIL choice\n"); ++failures; }

    // ---- helio_splat_bwd / helio_geometry_bwd / helio_render_bwd
    EXPECT_INVALID(helio_splat_bwd(B, N, 0, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_bwd(B, N, R, nullptr, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_bwd(B, N, R, P, P, P, nullptr, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_bwd(B, N, R, P, P, P, P, nullptr, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_bwd(B, N, R, P, P, P, P1, P, 0, nullptr, 0, nullptr));
    EXPECT_LAUNCH(helio_splat_bwd(B, N, R, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_splat_bwd(B, N, R, P, P, P, P, P, 0, P1, 1 << 20, nullptr));
    EXPECT_INVALID(helio_splat_bwd(B, N, R, P, P, P, P, P, 0, SCR, -1, nullptr));
    EXPECT_LAUNCH(helio_splat_bwd(512, 2000, 512, P, P, P, P, P, 2, SCR, 1l << 30, nullptr));
    EXPECT_LAUNCH(helio_splat_bwd(512, 2000, 512, P, P, P, P, P, 2, SCR, 8, nullptr));
    if (helio_splat_bwd_blocks(0) != 0 || helio_splat_bwd_blocks(129) < 1) { printf("FAIL bwd blocks\n"); ++failures; }
    EXPECT_INVALID(helio_geometry_bwd(B, N, 1, P, P, P, P, S, &plane, P, P, P, nullptr, nullptr));
    EXPECT_INVALID(helio_geometry_bwd(B, N, 0, P, P, P, P, S, &plane, P, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_bwd(B, N, 1, P, P, P, P, 5, &plane, P, P, P, P, nullptr));
    EXPECT_INVALID(helio_geometry_bwd(B, N, 1, P, P, P, P1, S, &plane, P, P, P, P, nullptr));
    EXPECT_LAUNCH(helio_geometry_bwd(B, N, 0, P, P, P, P, S, &plane, nullptr, P, nullptr, P, nullptr));
    EXPECT_INVALID(helio_render_bwd(B, N, R, P, P, P, P, S, &plane, P, P, P, P, P, P, nullptr, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_bwd(B, N, R, P, P, P, P, S, &plane, nullptr, P, P, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_bwd(B, N, R, P, P, P, P, S, &plane, P, P, P, P1, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_bwd(B, N, R, P, P, P, P, S, &plane, P, P, P, P, P, P, P, nullptr, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_bwd(B, 0, R, P, P, P, P, S, &plane, P, P, P, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_LAUNCH(helio_render_bwd(B, N, R, P, P, P, P, S, &plane, P, P, P, P, P, P, P, P, 0, nullptr, 0, nullptr));
    EXPECT_LAUNCH(helio_render_bwd(B, N, R, P, P, P, P, S, &plane, nullptr, nullptr, nullptr, nullptr, P, nullptr, nullptr, P, 0, nullptr, 0, nullptr));
    EXPECT_INVALID(helio_render_bwd(B, N, R, P, P, P, P, S, &plane, P, P, P, P, P, P, P, P, 0, P1, 64, nullptr));
    EXPECT_LAUNCH(helio_render_bwd(512, 2000, 512, P, P, P, P, 8000, &plane, P, P, P, P, P, P, P, P, 0, SCR, 1l << 30, nullptr));

    // ---- helio_ideal_normals / helio_distance_maps
    EXPECT_INVALID(helio_ideal_normals(B, N, nullptr, P, t3, P, nullptr));
    EXPECT_INVALID(helio_ideal_normals(B, N, P, P, nullptr, P, nullptr));
    EXPECT_INVALID(helio_ideal_normals(B, N, P, P, t3, nullptr, nullptr));
    EXPECT_INVALID(helio_ideal_normals(B, -3, P, P, t3, P, nullptr));
    EXPECT_LAUNCH(helio_ideal_normals(B, N, P, P, t3, P, nullptr));
    if (helio_distance_maps_workspace(0, R) != 0 || helio_distance_maps_workspace(B, 0) != 0 ||
        helio_distance_maps_workspace(B, R) != (long)B * R * R + 2 * B) { printf("FAIL edt workspace\n"); ++failures; }
    EXPECT_INVALID(helio_distance_maps(0, R, P, 0.5f, WS, P, nullptr));
    EXPECT_INVALID(helio_distance_maps(B, 20000, P, 0.5f, WS, P, nullptr));
    EXPECT_INVALID(helio_distance_maps(B, R, nullptr, 0.5f, WS, P, nullptr));
    EXPECT_INVALID(helio_distance_maps(B, R, P, 0.5f, nullptr, P, nullptr));
    EXPECT_INVALID(helio_distance_maps(B, R, P, 0.5f, WS, nullptr, nullptr));
    EXPECT_LAUNCH(helio_distance_maps(B, R, P, 0.5f, WS, P, nullptr));

    // ---- loss block
    if (helio_step_losses_workspace(0, N, R) != 0 || helio_step_losses_workspace(B, N, R) <= 0 ||
        helio_env_step_workspace(B, N, 0) != 0 || helio_env_step_workspace(B, N, R) <= 0) { printf("FAIL loss workspace\n"); ++failures; }
    EXPECT_INVALID(helio_step_losses_fwd(B, N, R, nullptr, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, -1.f, P, P, P, P, P, P, nullptr, nullptr, nullptr));
    EXPECT_INVALID(helio_step_losses_fwd(B, N, R, P, P, P, P, P, P, P, P, nullptr, n3, 15.f, 15.f, 0, -1.f, P, P, P, P, P, P, nullptr, nullptr, nullptr));
    EXPECT_INVALID(helio_step_losses_fwd(B, N, R, P, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, 1.5f, P, P, P, P, P, P, nullptr, nullptr, nullptr));
    EXPECT_INVALID(helio_step_losses_fwd(5000, N, R, P, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, 0.2f, P, P, P, P, P, P, nullptr, nullptr, nullptr));
    EXPECT_INVALID(helio_step_losses_fwd(B, N, R, P, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, -1.f, P, P, P, P, P, P, nullptr, P, nullptr));
    EXPECT_INVALID(helio_step_losses_fwd(B, N, R, P1, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, -1.f, P, P, P, P, P, P, nullptr, nullptr, nullptr));
    EXPECT_INVALID(helio_step_losses_fwd(B, N, R, P, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, -1.f, nullptr, P, P, P, P, P, nullptr, nullptr, nullptr));
    EXPECT_LAUNCH(helio_step_losses_fwd(B, N, R, P, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, 0.2f, P, P, P, P, P, P, P, P, nullptr));
    EXPECT_INVALID(helio_step_losses_bwd(B, N, R, P, nullptr, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, P, P, P, P, P, P, P, P, nullptr));
    EXPECT_INVALID(helio_step_losses_bwd(B, N, R, P, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, P, P, P, P, P, P1, P, P, nullptr));
    EXPECT_INVALID(helio_step_losses_bwd(B, N, 0, P, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, P, P, P, P, P, P, P, P, nullptr));
    EXPECT_LAUNCH(helio_step_losses_bwd(B, N, R, P, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, P, nullptr, nullptr, nullptr, nullptr, P, nullptr, nullptr, nullptr));

    // ---- env step (fused small problem and the composed large one)
#define STEP_FWD(Bv, Nv, Rv, helios, stride, ws, ratio, rec, ticket) STEP_FWD_S(Bv, Nv, Rv, helios, stride, ws, ratio, rec, ticket, nullptr, 0)
#define STEP_FWD_S(Bv, Nv, Rv, helios, stride, ws, ratio, rec, ticket, scratch, sbytes)                                                 \
    helio_env_step_fwd(Bv, Nv, Rv, helios, P, P, P, stride, &plane, P, P, P, P, P, P, 0, P, P, P, P, t3, n3, 15.f, 15.f, 0, \
                       ratio, ws, P, P, P, P, P, P, rec, ticket, scratch, sbytes, nullptr)
    EXPECT_INVALID(STEP_FWD(B, N, R, P, S, P, -1.f, REC, 0));                 // ticket 0 is reserved
    EXPECT_INVALID(STEP_FWD(B, N, R, nullptr, S, P, -1.f, nullptr, 0));
    EXPECT_INVALID(STEP_FWD(B, N, R, P, S + 4, P, -1.f, nullptr, 0));
    EXPECT_INVALID(STEP_FWD(B, N, R, P, S, nullptr, -1.f, nullptr, 0));
    EXPECT_INVALID(STEP_FWD(B, N, R, P, S, P, 2.0f, nullptr, 0));
    EXPECT_INVALID(STEP_FWD(0, N, R, P, S, P, -1.f, nullptr, 0));
    EXPECT_INVALID(STEP_FWD(B, N, 0, P, S, P, -1.f, nullptr, 0));
    EXPECT_LAUNCH(STEP_FWD(B, N, R, P, S, P, 0.2f, REC, 7));
    EXPECT_LAUNCH(STEP_FWD(512, 2000, 512, P, 8000, P, -1.f, nullptr, 0));
    EXPECT_INVALID(STEP_FWD_S(B, N, R, P, S, P, -1.f, nullptr, 0, P1, 64));
    EXPECT_INVALID(STEP_FWD_S(512, 2000, 512, P, 8000, P, -1.f, nullptr, 0, SCR, -1));
    EXPECT_LAUNCH(STEP_FWD_S(512, 2000, 512, P, 8000, P, -1.f, nullptr, 0, SCR, 1l << 30));
    if (helio_env_step_launches(B, N, R) != 2 || helio_env_step_launches(512, 2000, 512) != 4) { printf("FAIL step launches\n"); ++failures; }
#define STEP_BWD(Bv, Nv, Rv, stride, rays, gmse, gws, mom, grad) STEP_BWD_S(Bv, Nv, Rv, stride, rays, gmse, gws, mom, grad, nullptr, 0)
#define STEP_BWD_S(Bv, Nv, Rv, stride, rays, gmse, gws, mom, grad, scratch, sbytes)                                                     \
    helio_env_step_bwd(Bv, Nv, Rv, P, P, P, P, stride, &plane, rays, P, P, P, P, P, P, P, t3, n3, 15.f, 15.f, 0, gmse, \
                       nullptr, nullptr, P, nullptr, nullptr, nullptr, gws, mom, grad, 0, scratch, sbytes, nullptr)
    EXPECT_INVALID(STEP_BWD(B, N, R, S, P, P, P, P, nullptr));
    EXPECT_INVALID(STEP_BWD(B, N, R, S, nullptr, P, P, P, P));
    EXPECT_INVALID(STEP_BWD(B, N, R, S, P, P, P, nullptr, P));
    EXPECT_INVALID(STEP_BWD(B, N, R, 9, P, P, P, P, P));
    EXPECT_INVALID(STEP_BWD(40, 300, 256, 1200, P, P, nullptr, P, P));          // many rays: needs the image cotangent buffer
    EXPECT_INVALID(STEP_BWD(B, N, R, S, P1, P, P, P, P));
    EXPECT_INVALID(STEP_BWD(B, N, 0, S, P, P, P, P, P));
    EXPECT_LAUNCH(STEP_BWD(B, N, R, S, P, P, P, P, P));
    EXPECT_LAUNCH(STEP_BWD(40, 300, 256, 1200, P, P, P, P, P));
    EXPECT_LAUNCH(STEP_BWD(B, N, R, S, nullptr, nullptr, nullptr, nullptr, P));  // alignment only: no image path
    EXPECT_INVALID(STEP_BWD_S(B, N, R, S, P, P, P, P, P, P1, 64));
    EXPECT_LAUNCH(STEP_BWD_S(512, 2000, 512, 8000, P, P, P, P, P, SCR, 1l << 30));

    // ---- completion record
    EXPECT_INVALID(helio_notify_create(nullptr));
    EXPECT_INVALID(helio_notify_wait(nullptr, 3, 0.01));
    int slots[2 * 64];
    memset(slots, 0, sizeof slots);
    EXPECT_INVALID(helio_notify_wait(slots, 0, 0.01));
    if (helio_notify_wait(slots, 5, 0.01) != HELIO_E_TIMEOUT) { printf("FAIL notify timeout\n"); ++failures; }
    slots[2 * 5] = 1; slots[2 * 5 + 1] = 5;
    if (helio_notify_wait(slots, 5, 0.01) != 1) { printf("FAIL notify flag\n"); ++failures; }
    slots[2 * 5 + 1] = 5 + 64;                                                     // the slot has moved on
    if (helio_notify_wait(slots, 5, 0.01) != HELIO_E_STALE) { printf("FAIL notify stale\n"); ++failures; }
    if (helio_notify_destroy(nullptr) != HELIO_OK) { printf("FAIL notify destroy\n"); ++failures; }

    printf(failures ? "ABI SAN FAILED (%d)\n" : "ABI SAN OK\n", failures);
    return failures ? 1 : 0;
}
