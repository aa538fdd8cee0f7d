// extern "C" entry points of libhelio.so (declared in include/helio.h).
// Validates sizes and pointers on the host BEFORE any launch (a faulting kernel can
// reset the whole node), enqueues on the caller's stream, never synchronises, never
// allocates.  No exception crosses the boundary: errors are negative return codes plus
// a thread-local message.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "helio.h"

namespace helio {
void launch_geometry_fwd(int, int, const float*, const float*, const float*, const float*, long,
                         const helio_plane*, float*, float*, float*, hipStream_t);
void launch_geometry_bwd(int, int, int, const float*, const float*, const float*, const float*, long,
                         const helio_plane*, const float*, const float*, const float*, float*, hipStream_t);
void launch_ideal_normals(int, int, const float*, const float*, const float*, float*, hipStream_t);
void launch_error_trig(long, const float*, float*, hipStream_t);
void launch_init_actions(long, const float*, const float*, float, float*, hipStream_t);
int launch_splat_fwd(int, int, int, const float*, const float*, const float*, float*, int, void*, long, hipStream_t);
int launch_splat_bwd(int, int, int, const float*, const float*, const float*, const float*, float*, int, void*, long, hipStream_t);
long splat_fwd_scratch_bytes(int, int, int, int);
long splat_fwd_scratch_required(int, int, int, int);
long splat_bwd_scratch_bytes(int, int, int, int);
int splat_bwd_blocks(int);
bool render_is_fused(int, int, int);
bool render_is_fused_plain(int, int, int);
int render_fwd_choice(int, int, int);
void launch_distance_maps(int, int, const float*, float, int*, float*, int*, float*, hipStream_t);
int step_losses_chunks(int);
int step_losses_ray_wgs(int, int);
int step_losses_max_mask_batch();
void launch_step_losses_fwd(int, int, int, const float*, const float*, const float*, const float*, const float*,
                            const float*, const float*, const float*, const float*, const float*, float, float, int,
                            float, float*, float*, float*, float*, float*, float*, const float*, float*, int*, int,
                            hipStream_t);
void launch_step_losses_bwd(int, int, int, const float*, const float*, const float*, const float*, const float*,
                            const float*, const float*, const float*, const float*, const float*, float, float, int,
                            const float*, const float*, const float*, const float*, const float*, float*, float*,
                            float*, hipStream_t);
bool launch_render_fused(int, int, int, const float*, const float*, const float*, const float*, long,
                         const helio_plane*, const float*, const float*, float*, float*, float*, float*, int, hipStream_t);
bool splat_bwd_is_few(int, int);
void launch_splat_bwd_fused_loss_raw(int, int, int, const float*, const float*, const float*, const float*,
                                     const float*, const float*, const float*, const float*, const float*,
                                     const float*, float*, hipStream_t);
void launch_geometry_bwd_losses(int, int, int, const float*, const float*, const float*, const float*, long,
                                const helio_plane*, const float*, const float*, const float*, float*, const float*,
                                const float*, const float*, const float*, const float*, float, float, int,
                                hipStream_t);
long env_step_fused_workspace(int, int);
bool render_bwd_is_fused(int, int, int);
int render_bwd_choice(int, int, int);
bool launch_render_bwd_fused(int, int, int, const float*, const float*, const float*, const float*, const float*,
                             const float*, const float*, const float*, long, const helio_plane*, const float*,
                             const float*, float*, const float*, const float*, const float*, const float*,
                             const float*, float, float, int, hipStream_t);
bool launch_env_step_fused(int, int, int, const float*, const float*, const float*, const float*, long,
                           const helio_plane*, const float*, const float*, float*, float*, float*, float*,
                           const float*, const float*, const float*, const float*, const float*, const float*, float,
                           float, int, float, float*, float*, float*, float*, float*, float*, float*, int*, int, int,
                           hipStream_t);
}  // namespace helio

namespace {
thread_local char g_err[256] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int after_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(HELIO_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return HELIO_OK;
}

// variants 10..13 force one form of the single-launch render (parity tests, tuning): the block kernel
// with 1 / 2 / 4 waves per block, or the few-ray streaming kernel; → 0 when the variant is not one of them
int fused_form(int variant) { return variant == 10 ? 1 : variant == 11 ? 2 : variant == 12 ? 4 : variant == 13 ? 8 : 0; }

bool sizes_ok(int B, int N) { return B >= 1 && N >= 1 && B <= 65535 && (long)B * N <= (1l << 31) / 4; }
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// optional device scratch: NULL (with any size) or 256-byte aligned with a non-negative size
bool scratch_ok(const void* p, long bytes) { return !p || ((reinterpret_cast<uintptr_t>(p) & 255) == 0 && bytes >= 0); }
}  // namespace

extern "C" {

int helio_abi_version(void) { return HELIO_ABI_VERSION; }
const char* helio_last_error_string(void) { return g_err; }

int helio_device_arch(int device, char* buf, int buflen) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(HELIO_E_NODEVICE, "no HIP device %d", device);
    }
    if (buf && buflen > 0) { strncpy(buf, prop.gcnArchName, buflen - 1); buf[buflen - 1] = 0; }
    return HELIO_OK;
}

int helio_geometry_fwd(int B, int N, const float* helios_d, const float* sun_d, const float* action_d,
                       const float* trig_d, long trig_b_stride, const helio_plane* plane,
                       float* actual_d, float* refl_d, float* rays_d, void* stream) {
    if (!sizes_ok(B, N)) return fail(HELIO_E_INVALID, "geometry_fwd: bad sizes B=%d N=%d", B, N);
    if (!helios_d || !sun_d || !action_d || !trig_d || !plane || !actual_d)
        return fail(HELIO_E_INVALID, "geometry_fwd: null pointer");
    if (trig_b_stride != 0 && trig_b_stride != 4l * N)
        return fail(HELIO_E_INVALID, "geometry_fwd: trig_b_stride must be 0 or 4*N");
    if (!aligned16(trig_d) || (rays_d && !aligned16(rays_d)))
        return fail(HELIO_E_INVALID, "geometry_fwd: trig/rays must be 16-byte aligned");
    helio::launch_geometry_fwd(B, N, helios_d, sun_d, action_d, trig_d, trig_b_stride, plane, actual_d,
                               refl_d, rays_d, static_cast<hipStream_t>(stream));
    return after_launch("geometry_fwd");
}

int helio_splat_fwd(int B, int N, int R, const float* rays_d, const float* xs_d, const float* ys_d,
                    float* image_d, int variant, void* scratch_d, long scratch_bytes, void* stream) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return fail(HELIO_E_INVALID, "splat_fwd: bad sizes B=%d N=%d R=%d", B, N, R);
    if (!rays_d || !xs_d || !ys_d || !image_d) return fail(HELIO_E_INVALID, "splat_fwd: null pointer");
    if (!aligned16(rays_d) || !aligned16(image_d)) return fail(HELIO_E_INVALID, "splat_fwd: rays/image must be 16-byte aligned");
    if (!scratch_ok(scratch_d, scratch_bytes)) return fail(HELIO_E_INVALID, "splat_fwd: scratch must be 256-byte aligned");
    if (const int rc = helio::launch_splat_fwd(B, N, R, rays_d, xs_d, ys_d, image_d, variant, scratch_d, scratch_bytes,
                                               static_cast<hipStream_t>(stream)); rc != HELIO_OK)
        return rc == HELIO_E_SCRATCH ? fail(rc, "splat_fwd: variant %d at B=%d N=%d R=%d needs %ld bytes of scratch", variant, B, N, R,
                                            helio::splat_fwd_scratch_required(B, N, R, variant))
                                     : fail(HELIO_E_INVALID, "splat_fwd: unknown variant %d", variant);
    return after_launch("splat_fwd");
}

int helio_render_fwd(int B, int N, int R, const float* helios_d, const float* sun_d, const float* action_d,
                     const float* trig_d, long trig_b_stride, const helio_plane* plane, const float* xs_d,
                     const float* ys_d, float* actual_d, float* refl_d, float* rays_d, float* image_d,
                     int variant, void* scratch_d, long scratch_bytes, void* stream) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return fail(HELIO_E_INVALID, "render_fwd: bad sizes B=%d N=%d R=%d", B, N, R);
    if (!helios_d || !sun_d || !action_d || !trig_d || !plane || !xs_d || !ys_d || !actual_d || !image_d)
        return fail(HELIO_E_INVALID, "render_fwd: null pointer");
    if (trig_b_stride != 0 && trig_b_stride != 4l * N)
        return fail(HELIO_E_INVALID, "render_fwd: trig_b_stride must be 0 or 4*N");
    if (!aligned16(trig_d) || (rays_d && !aligned16(rays_d)) || !aligned16(image_d))
        return fail(HELIO_E_INVALID, "render_fwd: trig/rays/image must be 16-byte aligned");
    if (!scratch_ok(scratch_d, scratch_bytes)) return fail(HELIO_E_INVALID, "render_fwd: scratch must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (((variant == 0 || variant == 2) && helio::render_is_fused_plain(B, N, R)) || fused_form(variant)) {
        if (!helio::launch_render_fused(B, N, R, helios_d, sun_d, action_d, trig_d, trig_b_stride, plane, xs_d, ys_d,
                                        actual_d, refl_d, rays_d, image_d, fused_form(variant), st))
            return fail(HELIO_E_INVALID, "render_fwd: variant %d does not exist for B=%d N=%d R=%d", variant, B, N, R);
        return after_launch("render_fwd(fused)");
    }
    if (!rays_d) return fail(HELIO_E_INVALID, "render_fwd: this problem size needs the rays work buffer");
    if (const long need = helio::splat_fwd_scratch_required(B, N, R, variant); need > 0 && (!scratch_d || scratch_bytes < need))
        return fail(HELIO_E_SCRATCH, "render_fwd: variant %d at B=%d N=%d R=%d needs %ld bytes of scratch", variant, B, N, R, need);
    helio::launch_geometry_fwd(B, N, helios_d, sun_d, action_d, trig_d, trig_b_stride, plane, actual_d, refl_d, rays_d, st);
    if (const int rc = helio::launch_splat_fwd(B, N, R, rays_d, xs_d, ys_d, image_d, variant, scratch_d, scratch_bytes, st);
        rc != HELIO_OK)
        return rc == HELIO_E_SCRATCH ? fail(rc, "render_fwd: variant %d at B=%d N=%d R=%d needs %ld bytes of scratch", variant, B, N, R,
                                            helio::splat_fwd_scratch_required(B, N, R, variant))
                                     : fail(HELIO_E_INVALID, "render_fwd: unknown variant %d", variant);
    return after_launch("render_fwd");
}

int helio_render_bwd_choice(int B, int N, int R) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return 0;
    return helio::render_bwd_choice(B, N, R);
}

int helio_render_fwd_launches(int B, int N, int R) {
    return (sizes_ok(B, N) && R >= 1 && helio::render_is_fused_plain(B, N, R)) ? 1 : 2;
}

long helio_fwd_scratch_bytes(int B, int N, int R, int variant) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return 0;
    if (((variant == 0 || variant == 2) && helio::render_is_fused_plain(B, N, R)) || fused_form(variant)) return 0;
    return helio::splat_fwd_scratch_bytes(B, N, R, variant);
}

long helio_fwd_scratch_required(int B, int N, int R, int variant) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return 0;
    if (((variant == 0 || variant == 2) && helio::render_is_fused_plain(B, N, R)) || fused_form(variant)) return 0;
    return helio::splat_fwd_scratch_required(B, N, R, variant);
}

long helio_bwd_scratch_bytes(int B, int N, int R, int variant) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return 0;
    if (variant == 8 || (variant == 0 && helio::render_bwd_is_fused(B, N, R))) return 0;
    return helio::splat_bwd_scratch_bytes(B, N, R, variant);
}

int helio_splat_bwd_blocks(int R) { return R >= 1 ? helio::splat_bwd_blocks(R) : 0; }

int helio_splat_bwd(int B, int N, int R, const float* rays_d, const float* xs_d, const float* ys_d,
                    const float* grad_image_d, float* moments_d, int variant, void* scratch_d, long scratch_bytes,
                    void* stream) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return fail(HELIO_E_INVALID, "splat_bwd: bad sizes B=%d N=%d R=%d", B, N, R);
    if (!rays_d || !xs_d || !ys_d || !grad_image_d || !moments_d) return fail(HELIO_E_INVALID, "splat_bwd: null pointer");
    if (!aligned16(rays_d) || !aligned16(grad_image_d)) return fail(HELIO_E_INVALID, "splat_bwd: rays/grad_image must be 16-byte aligned");
    if (!scratch_ok(scratch_d, scratch_bytes)) return fail(HELIO_E_INVALID, "splat_bwd: scratch must be 256-byte aligned");
    if (helio::launch_splat_bwd(B, N, R, rays_d, xs_d, ys_d, grad_image_d, moments_d, variant, scratch_d, scratch_bytes,
                                static_cast<hipStream_t>(stream)) != HELIO_OK)
        return fail(HELIO_E_INVALID, "splat_bwd: unknown variant %d", variant);
    return after_launch("splat_bwd");
}

int helio_geometry_bwd(int B, int N, int n_blocks, const float* helios_d, const float* sun_d,
                       const float* action_d, const float* trig_d, long trig_b_stride,
                       const helio_plane* plane, const float* moments_d, const float* grad_actual_d,
                       const float* grad_refl_d, float* grad_action_d, void* stream) {
    if (!sizes_ok(B, N)) return fail(HELIO_E_INVALID, "geometry_bwd: bad sizes B=%d N=%d", B, N);
    if (!helios_d || !sun_d || !action_d || !trig_d || !plane || !grad_action_d)
        return fail(HELIO_E_INVALID, "geometry_bwd: null pointer");
    if (moments_d && n_blocks < 1) return fail(HELIO_E_INVALID, "geometry_bwd: n_blocks must be >= 1 with moments");
    if (trig_b_stride != 0 && trig_b_stride != 4l * N)
        return fail(HELIO_E_INVALID, "geometry_bwd: trig_b_stride must be 0 or 4*N");
    if (!aligned16(trig_d)) return fail(HELIO_E_INVALID, "geometry_bwd: trig must be 16-byte aligned");
    helio::launch_geometry_bwd(B, N, n_blocks, helios_d, sun_d, action_d, trig_d, trig_b_stride, plane,
                               moments_d, grad_actual_d, grad_refl_d, grad_action_d,
                               static_cast<hipStream_t>(stream));
    return after_launch("geometry_bwd");
}

int helio_render_bwd(int B, int N, int R, const float* helios_d, const float* sun_d, const float* action_d,
                     const float* trig_d, long trig_b_stride, const helio_plane* plane, const float* rays_d,
                     const float* xs_d, const float* ys_d, const float* grad_image_d, const float* grad_actual_d,
                     const float* grad_refl_d, float* moments_d, float* grad_action_d, int variant, void* scratch_d,
                     long scratch_bytes, void* stream) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return fail(HELIO_E_INVALID, "render_bwd: bad sizes B=%d N=%d R=%d", B, N, R);
    if (!helios_d || !sun_d || !action_d || !trig_d || !plane || !grad_action_d)
        return fail(HELIO_E_INVALID, "render_bwd: null pointer");
    if (trig_b_stride != 0 && trig_b_stride != 4l * N)
        return fail(HELIO_E_INVALID, "render_bwd: trig_b_stride must be 0 or 4*N");
    if (!aligned16(trig_d)) return fail(HELIO_E_INVALID, "render_bwd: trig must be 16-byte aligned");
    if (!scratch_ok(scratch_d, scratch_bytes)) return fail(HELIO_E_INVALID, "render_bwd: scratch must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (grad_image_d) {
        if (!rays_d || !xs_d || !ys_d || !moments_d) return fail(HELIO_E_INVALID, "render_bwd: null pointer");
        if (!aligned16(rays_d) || !aligned16(grad_image_d))
            return fail(HELIO_E_INVALID, "render_bwd: rays/grad_image must be 16-byte aligned");
        // variant 8 forces, variant 0 chooses by size: moments and the geometry adjoint in ONE launch
        if (variant == 8 || (variant == 0 && helio::render_bwd_is_fused(B, N, R))) {
            if (!helio::launch_render_bwd_fused(B, N, R, rays_d, xs_d, ys_d, grad_image_d, helios_d, sun_d, action_d, trig_d,
                                                trig_b_stride, plane, grad_actual_d, grad_refl_d, grad_action_d, nullptr,
                                                nullptr, nullptr, nullptr, nullptr, 0.0f, 0.0f, 0, st))
                return fail(HELIO_E_INVALID, "render_bwd: variant 8 does not exist for B=%d N=%d R=%d", B, N, R);
            return after_launch("render_bwd");
        }
        if (helio::launch_splat_bwd(B, N, R, rays_d, xs_d, ys_d, grad_image_d, moments_d, variant, scratch_d, scratch_bytes, st) != HELIO_OK)
            return fail(HELIO_E_INVALID, "render_bwd: unknown variant %d", variant);
    }
    helio::launch_geometry_bwd(B, N, helio::splat_bwd_blocks(R), helios_d, sun_d, action_d, trig_d, trig_b_stride,
                               plane, grad_image_d ? moments_d : nullptr, grad_actual_d, grad_refl_d, grad_action_d, st);
    return after_launch("render_bwd");
}

int helio_error_trig(long M, const float* errs_d, float* trig_d, void* stream) {
    if (M < 1 || M > (1l << 31) / 4) return fail(HELIO_E_INVALID, "error_trig: bad size M=%ld", M);
    if (!errs_d || !trig_d) return fail(HELIO_E_INVALID, "error_trig: null pointer");
    if ((reinterpret_cast<uintptr_t>(errs_d) & 7) || !aligned16(trig_d))
        return fail(HELIO_E_INVALID, "error_trig: errs must be 8-byte and trig 16-byte aligned");
    helio::launch_error_trig(M, errs_d, trig_d, static_cast<hipStream_t>(stream));
    return after_launch("error_trig");
}

int helio_ideal_normals(int B, int N, const float* helios_d, const float* sun_d,
                        const float target_position[3], float* out_d, void* stream) {
    if (!sizes_ok(B, N)) return fail(HELIO_E_INVALID, "ideal_normals: bad sizes B=%d N=%d", B, N);
    if (!helios_d || !sun_d || !target_position || !out_d) return fail(HELIO_E_INVALID, "ideal_normals: null pointer");
    helio::launch_ideal_normals(B, N, helios_d, sun_d, target_position, out_d, static_cast<hipStream_t>(stream));
    return after_launch("ideal_normals");
}

int helio_init_actions(long M, const float* ideal_d, const float* noise_d, float noise_scale, float* out_d,
                       void* stream) {
    if (M < 1 || M > (1l << 31) / 4) return fail(HELIO_E_INVALID, "init_actions: bad size M=%ld", M);
    if (!ideal_d || !noise_d || !out_d) return fail(HELIO_E_INVALID, "init_actions: null pointer");
    helio::launch_init_actions(M, ideal_d, noise_d, noise_scale, out_d, static_cast<hipStream_t>(stream));
    return after_launch("init_actions");
}

long helio_distance_maps_workspace(int B, int R) {
    if (B < 1 || R < 1 || R > 16384) return 0;
    return (long)B * R * R + 2l * B;                 // int g[B,R,R], float max[B], int any_hot[B]
}

int helio_distance_maps(int B, int R, const float* img_d, float thr, void* workspace_d, float* out_d, void* stream) {
    if (B < 1 || B > 65535 || R < 1 || R > 16384) return fail(HELIO_E_INVALID, "distance_maps: bad sizes B=%d R=%d", B, R);
    if (!img_d || !workspace_d || !out_d) return fail(HELIO_E_INVALID, "distance_maps: null pointer");
    int* g = static_cast<int*>(workspace_d);
    float* mx = reinterpret_cast<float*>(g + (long)B * R * R);
    int* any_hot = reinterpret_cast<int*>(mx + B);
    helio::launch_distance_maps(B, R, img_d, thr, g, mx, any_hot, out_d, static_cast<hipStream_t>(stream));
    return after_launch("distance_maps");
}

long helio_step_losses_workspace(int B, int N, int R) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return 0;
    return 3l * B * helio::step_losses_chunks(R) + 2l * helio::step_losses_ray_wgs(B, N);
}

int helio_step_losses_fwd(int B, int N, int R, const float* img_d, const float* target_d, const float* tx_d,
                          const float* dmaps_d, const float* ideal_d, const float* actual_d, const float* action_d,
                          const float* helios_d, const float target_position[3], const float target_normal[3],
                          float width, float height, int exponential_risk, float error_mask_ratio,
                          float* workspace_d, float* out_d, float* mae_d, float* keep_d, float* align_err_d,
                          float* all_bounds_d, const float* sun_d, float* aux_d, void* stream) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return fail(HELIO_E_INVALID, "step_losses_fwd: bad sizes B=%d N=%d R=%d", B, N, R);
    if (!img_d || !target_d || !tx_d || !dmaps_d || !ideal_d || !actual_d || !action_d || !helios_d ||
        !target_position || !target_normal || !workspace_d || !out_d || !mae_d || !keep_d || !align_err_d ||
        !all_bounds_d)
        return fail(HELIO_E_INVALID, "step_losses_fwd: null pointer");
    if (aux_d && !sun_d) return fail(HELIO_E_INVALID, "step_losses_fwd: aux needs the sun positions");
    if (error_mask_ratio >= 0.0f && (error_mask_ratio > 1.0f || B > helio::step_losses_max_mask_batch()))
        return fail(HELIO_E_INVALID, "step_losses_fwd: error mask needs ratio in [0,1] and B <= %d",
                    helio::step_losses_max_mask_batch());
    if (!aligned16(img_d) || !aligned16(target_d) || !aligned16(dmaps_d))
        return fail(HELIO_E_INVALID, "step_losses_fwd: images must be 16-byte aligned");
    helio::launch_step_losses_fwd(B, N, R, img_d, target_d, tx_d, dmaps_d, ideal_d, actual_d, action_d, helios_d,
                                  target_position, target_normal, width, height, exponential_risk, error_mask_ratio,
                                  workspace_d, out_d, mae_d, keep_d, align_err_d, all_bounds_d, sun_d, aux_d,
                                  nullptr, 0, static_cast<hipStream_t>(stream));
    return after_launch("step_losses_fwd");
}

int helio_step_losses_bwd(int B, int N, int R, const float* img_d, const float* target_d, const float* tx_d,
                          const float* dmaps_d, const float* ideal_d, const float* actual_d, const float* action_d,
                          const float* helios_d, const float target_position[3], const float target_normal[3],
                          float width, float height, int exponential_risk, const float* g_mse_d,
                          const float* g_dist_d, const float* g_bound_d, const float* g_align_d, const float* keep_d,
                          float* grad_img_d, float* grad_actual_d, float* grad_action_d, void* stream) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return fail(HELIO_E_INVALID, "step_losses_bwd: bad sizes B=%d N=%d R=%d", B, N, R);
    if (!img_d || !target_d || !tx_d || !dmaps_d || !ideal_d || !actual_d || !action_d || !helios_d ||
        !target_position || !target_normal)
        return fail(HELIO_E_INVALID, "step_losses_bwd: null pointer");
    if (!aligned16(img_d) || !aligned16(target_d) || !aligned16(dmaps_d) || (grad_img_d && !aligned16(grad_img_d)))
        return fail(HELIO_E_INVALID, "step_losses_bwd: images must be 16-byte aligned");
    helio::launch_step_losses_bwd(B, N, R, img_d, target_d, tx_d, dmaps_d, ideal_d, actual_d, action_d, helios_d,
                                  target_position, target_normal, width, height, exponential_risk, g_mse_d, g_dist_d,
                                  g_bound_d, g_align_d, keep_d, grad_img_d, grad_actual_d, grad_action_d,
                                  static_cast<hipStream_t>(stream));
    return after_launch("step_losses_bwd");
}

long helio_env_step_workspace(int B, int N, int R) {
    const long a = helio_step_losses_workspace(B, N, R);
    if (a == 0) return 0;
    const long f = helio::env_step_fused_workspace(B, R);
    return a > f ? a : f;
}

int helio_render_fwd_choice(int B, int N, int R) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return 0;
    return helio::render_fwd_choice(B, N, R);
}

int helio_env_step_launches(int B, int N, int R) {
    return (sizes_ok(B, N) && R >= 1 && helio::render_is_fused(B, N, R)) ? 2 : 4;
}

int helio_env_step_fwd(int B, int N, int R, const float* helios_d, const float* sun_d, const float* action_d,
                       const float* trig_d, long trig_b_stride, const helio_plane* plane, const float* xs_d,
                       const float* ys_d, float* actual_d, float* refl_d, float* rays_d, float* image_d, int variant,
                       const float* target_d, const float* tx_d, const float* dmaps_d, const float* ideal_d,
                       const float target_position[3], const float target_normal[3], float width, float height,
                       int exponential_risk, float error_mask_ratio, float* workspace_d, float* out_d, float* mae_d,
                       float* keep_d, float* align_err_d, float* all_bounds_d, float* aux_d, int* notify, int ticket,
                       void* scratch_d, long scratch_bytes, void* stream) {
    if (notify && ticket == 0) return fail(HELIO_E_INVALID, "env_step_fwd: ticket 0 is reserved");
    if (!scratch_ok(scratch_d, scratch_bytes)) return fail(HELIO_E_INVALID, "env_step_fwd: scratch must be 256-byte aligned");
    if (sizes_ok(B, N) && R >= 1 && R <= 16384 &&
        (((variant == 0 || variant == 2) && helio::render_is_fused(B, N, R)) || fused_form(variant))) {
        if (!helios_d || !sun_d || !action_d || !trig_d || !plane || !xs_d || !ys_d || !actual_d || !image_d ||
            !target_d || !tx_d || !dmaps_d || !ideal_d || !target_position || !target_normal || !workspace_d ||
            !out_d || !mae_d || !keep_d || !align_err_d || !all_bounds_d)
            return fail(HELIO_E_INVALID, "env_step_fwd: null pointer");
        if (trig_b_stride != 0 && trig_b_stride != 4l * N)
            return fail(HELIO_E_INVALID, "env_step_fwd: trig_b_stride must be 0 or 4*N");
        if (!aligned16(trig_d) || (rays_d && !aligned16(rays_d)))
            return fail(HELIO_E_INVALID, "env_step_fwd: trig/rays must be 16-byte aligned");
        if (error_mask_ratio >= 0.0f && (error_mask_ratio > 1.0f || B > helio::step_losses_max_mask_batch()))
            return fail(HELIO_E_INVALID, "env_step_fwd: error mask needs ratio in [0,1] and B <= %d",
                        helio::step_losses_max_mask_batch());
        if (!helio::launch_env_step_fused(B, N, R, helios_d, sun_d, action_d, trig_d, trig_b_stride, plane, xs_d, ys_d,
                                          actual_d, refl_d, rays_d, image_d, target_d, tx_d, dmaps_d, ideal_d,
                                          target_position, target_normal, width, height, exponential_risk,
                                          error_mask_ratio, workspace_d, out_d, mae_d, keep_d, align_err_d,
                                          all_bounds_d, aux_d, notify, ticket, fused_form(variant),
                                          static_cast<hipStream_t>(stream)))
            return fail(HELIO_E_INVALID, "env_step_fwd: variant %d does not exist for B=%d N=%d R=%d", variant, B, N, R);
        return after_launch("env_step_fwd(fused)");
    }
    const int rc = helio_render_fwd(B, N, R, helios_d, sun_d, action_d, trig_d, trig_b_stride, plane, xs_d, ys_d,
                                    actual_d, refl_d, rays_d, image_d, variant, scratch_d, scratch_bytes, stream);
    if (rc != HELIO_OK) return rc;
    // the two-call form, with the same checks as helio_step_losses_fwd
    if (!target_d || !tx_d || !dmaps_d || !ideal_d || !target_position || !target_normal || !workspace_d || !out_d ||
        !mae_d || !keep_d || !align_err_d || !all_bounds_d)
        return fail(HELIO_E_INVALID, "env_step_fwd: null pointer");
    if (error_mask_ratio >= 0.0f && (error_mask_ratio > 1.0f || B > helio::step_losses_max_mask_batch()))
        return fail(HELIO_E_INVALID, "env_step_fwd: error mask needs ratio in [0,1] and B <= %d",
                    helio::step_losses_max_mask_batch());
    if (!aligned16(target_d) || !aligned16(dmaps_d))
        return fail(HELIO_E_INVALID, "env_step_fwd: images must be 16-byte aligned");
    helio::launch_step_losses_fwd(B, N, R, image_d, target_d, tx_d, dmaps_d, ideal_d, actual_d, action_d, helios_d,
                                  target_position, target_normal, width, height, exponential_risk, error_mask_ratio,
                                  workspace_d, out_d, mae_d, keep_d, align_err_d, all_bounds_d, sun_d, aux_d, notify,
                                  ticket, static_cast<hipStream_t>(stream));
    return after_launch("env_step_fwd");
}

int helio_notify_create(int** record) {
    if (!record) return fail(HELIO_E_INVALID, "notify_create: null pointer");
    void* p = nullptr;
    if (hipHostMalloc(&p, 2 * HELIO_NOTIFY_SLOTS * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return fail(HELIO_E_NODEVICE, "notify_create: hipHostMalloc failed");
    }
    memset(p, 0, 2 * HELIO_NOTIFY_SLOTS * sizeof(int));
    *record = static_cast<int*>(p);
    return HELIO_OK;
}

int helio_notify_destroy(int* record) {
    if (record && hipHostFree(record) != hipSuccess) { (void)hipGetLastError(); return fail(HELIO_E_INVALID, "notify_destroy: not a record"); }
    return HELIO_OK;
}

int helio_notify_wait(const int* record, int ticket, double timeout_seconds) {
    if (!record || ticket == 0) return fail(HELIO_E_INVALID, "notify_wait: null record or ticket 0");
    const int* slot = record + 2 * (ticket & (HELIO_NOTIFY_SLOTS - 1));
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 1;; ++spin) {
        const int seen = __atomic_load_n(slot + 1, __ATOMIC_ACQUIRE);
        if (seen == ticket) return __atomic_load_n(slot, __ATOMIC_RELAXED) ? 1 : 0;
        if (seen != 0 && (int)((unsigned)seen - (unsigned)ticket) > 0)
            return fail(HELIO_E_STALE, "notify_wait: slot reused by ticket %d while waiting for %d", seen, ticket);
        __builtin_ia32_pause();
        if ((spin & 1023) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds)
            return fail(HELIO_E_TIMEOUT, "notify_wait: ticket %d not published within %.3f s", ticket, timeout_seconds);
    }
}

int helio_env_step_bwd_image_ws(int B, int N, int R) {
    (void)R;
    return helio::splat_bwd_is_few(B, N) ? 0 : 1;
}

int helio_env_step_bwd(int B, int N, int R, const float* helios_d, const float* sun_d, const float* action_d,
                       const float* trig_d, long trig_b_stride, const helio_plane* plane, const float* rays_d,
                       const float* xs_d, const float* ys_d, const float* image_d, const float* target_d,
                       const float* tx_d, const float* dmaps_d, const float* ideal_d, const float target_position[3],
                       const float target_normal[3], float width, float height, int exponential_risk,
                       const float* g_mse_d, const float* g_dist_d, const float* g_bound_d, const float* g_align_d,
                       const float* keep_d, const float* grad_actual_d, const float* grad_refl_d,
                       float* grad_image_ws_d, float* moments_d, float* grad_action_d, int variant, void* scratch_d,
                       long scratch_bytes, void* stream) {
    if (!sizes_ok(B, N) || R < 1 || R > 16384) return fail(HELIO_E_INVALID, "env_step_bwd: bad sizes B=%d N=%d R=%d", B, N, R);
    if (!helios_d || !sun_d || !action_d || !trig_d || !plane || !ideal_d || !target_position || !target_normal ||
        !grad_action_d)
        return fail(HELIO_E_INVALID, "env_step_bwd: null pointer");
    if (trig_b_stride != 0 && trig_b_stride != 4l * N)
        return fail(HELIO_E_INVALID, "env_step_bwd: trig_b_stride must be 0 or 4*N");
    if (!aligned16(trig_d)) return fail(HELIO_E_INVALID, "env_step_bwd: trig must be 16-byte aligned");
    if (!scratch_ok(scratch_d, scratch_bytes)) return fail(HELIO_E_INVALID, "env_step_bwd: scratch must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool through_image = g_mse_d || g_dist_d;
    if (through_image) {
        if (!rays_d || !xs_d || !ys_d || !image_d || !target_d || !tx_d || !dmaps_d || !moments_d)
            return fail(HELIO_E_INVALID, "env_step_bwd: null pointer (image path)");
        if (!aligned16(rays_d)) return fail(HELIO_E_INVALID, "env_step_bwd: rays must be 16-byte aligned");
        if ((variant == 0 || variant == 4) && helio::splat_bwd_is_few(B, N)) {
            helio::launch_splat_bwd_fused_loss_raw(B, N, R, rays_d, xs_d, ys_d, image_d, target_d, tx_d, dmaps_d,
                                                   keep_d, g_mse_d, g_dist_d, moments_d, st);
        } else {
            if (!grad_image_ws_d) return fail(HELIO_E_INVALID, "env_step_bwd: this problem size needs grad_image_ws_d");
            if (!aligned16(image_d) || !aligned16(target_d) || !aligned16(dmaps_d) || !aligned16(grad_image_ws_d))
                return fail(HELIO_E_INVALID, "env_step_bwd: images must be 16-byte aligned");
            helio::launch_step_losses_bwd(B, N, R, image_d, target_d, tx_d, dmaps_d, ideal_d, nullptr, action_d,
                                          helios_d, target_position, target_normal, width, height, exponential_risk,
                                          g_mse_d, g_dist_d, nullptr, nullptr, keep_d, grad_image_ws_d, nullptr,
                                          nullptr, st);
            if (variant == 8 || (variant == 0 && helio::render_bwd_is_fused(B, N, R))) {
                const bool ray_losses = g_align_d || g_bound_d;
                if (!helio::launch_render_bwd_fused(B, N, R, rays_d, xs_d, ys_d, grad_image_ws_d, helios_d, sun_d, action_d,
                                                    trig_d, trig_b_stride, plane, grad_actual_d, grad_refl_d, grad_action_d,
                                                    ray_losses ? ideal_d : nullptr, g_align_d, g_bound_d, target_position,
                                                    target_normal, width, height, exponential_risk, st))
                    return fail(HELIO_E_INVALID, "env_step_bwd: variant 8 does not exist for B=%d N=%d R=%d", B, N, R);
                return after_launch("env_step_bwd");
            }
            if (helio::launch_splat_bwd(B, N, R, rays_d, xs_d, ys_d, grad_image_ws_d, moments_d, variant, scratch_d, scratch_bytes, st) != HELIO_OK)
                return fail(HELIO_E_INVALID, "env_step_bwd: unknown variant %d", variant);
        }
    }
    helio::launch_geometry_bwd_losses(B, N, helio::splat_bwd_blocks(R), helios_d, sun_d, action_d, trig_d,
                                      trig_b_stride, plane, through_image ? moments_d : nullptr, grad_actual_d,
                                      grad_refl_d, grad_action_d, (g_align_d || g_bound_d) ? ideal_d : nullptr,
                                      g_align_d, g_bound_d, target_position, target_normal, width, height,
                                      exponential_risk, st);
    return after_launch("env_step_bwd");
}

}  // extern "C"
