/*
 * helio.h — C ABI of libhelio.so: the MI355X (gfx950) implementation of DOODLE's
 * differentiable heliostat render hot path.
 *
 * The reference (github.com/l3th4l/DOODLE) is pure Python/PyTorch and has NO
 * FFI/plugin interface of its own; its boundary for this path is the Python
 * class surface HelioField.render / calculate_ideal_normals / init_actions
 * (newenv_rl_test_multi_error.py:256-415).  The entry points below are what a
 * binding for that surface calls; doodle_amd/native.py binds them with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 * "Replaces" comments cite reference file:line in newenv_rl_test_multi_error.py.
 *
 * Conventions
 *   - every pointer named *_d is DEVICE memory owned by the caller (a torch
 *     tensor); fp32, dense row-major with the stated shape.  The library never
 *     allocates, frees or keeps a pointer after returning.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no
 *     call synchronises.
 *   - return value: 0 on success, a negative HELIO_E_* code otherwise;
 *     helio_last_error_string() describes the last failure on this thread.
 *   - B = sun positions, N = heliostats, R = receiver resolution (image R x R).
 */
#ifndef HELIO_H
#define HELIO_H

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped when a signature or the meaning of an argument changes.  Additions that leave every existing call
 * valid do not bump it.  2: the footprint entry points (helio_splat_fwd/bwd, helio_render_fwd/bwd,
 * helio_env_step_fwd/bwd) take an optional device scratch buffer (scratch_d, scratch_bytes) in front of the
 * stream — see "Device scratch" below; helio_fwd_scratch_bytes / helio_bwd_scratch_bytes size it. */
#define HELIO_ABI_VERSION 2

#define HELIO_OK            0
#define HELIO_E_INVALID    -1   /* bad size / null pointer                       */
#define HELIO_E_LAUNCH     -2   /* hipGetLastError() after a launch was not ok   */
#define HELIO_E_NODEVICE   -3   /* no HIP device / wrong architecture            */
#define HELIO_E_TIMEOUT    -4   /* helio_notify_wait: the record was not written in time */
#define HELIO_E_STALE      -5   /* helio_notify_wait: the slot was reused by a later ticket */
#define HELIO_E_SCRATCH    -6   /* the kernel this call runs NEEDS device scratch (helio_fwd_scratch_required) */

/* Receiver plane, HOST memory (passed by value to the kernels).
 * origin = target_position, normal = unit target normal (ctor :184-192),
 * u/v = plane_u/plane_v (:206-213), w = u x v (unit; the image plane's normal),
 * sigma_scale (:197). */
typedef struct helio_plane {
    float origin[3];
    float normal[3];
    float u[3];
    float v[3];
    float w[3];
    float sigma_scale;
} helio_plane;

/* Number of floats per ray in the `rays` work buffer written by
 * helio_geometry_fwd and read by the splat kernels:
 *   rays[b,n,:] = (a, b, k2, c2)  with
 *   gauss_bn[i,j] = exp2( -((xs[i]+a)^2 + (ys[j]+b)^2 + c2) * k2 ),
 *   k2 = log2(e)/max(2 sigma^2,1e-12) for a valid ray and 0 for a plane-parallel
 *   one (which then contributes exactly 1.0 to every pixel, as :141-148 does). */
#define HELIO_RAY_STRIDE 4
/* Floats per ray and per column block in the moment buffer (splat backward). */
#define HELIO_MOMENT_STRIDE 5

/*
 * Device scratch (optional, every footprint entry point): `scratch_d` is caller-owned device memory of
 * `scratch_bytes` bytes, 256-byte aligned, free for the library to overwrite during the call and of no meaning
 * afterwards; NULL = none.  With at least helio_fwd_scratch_bytes(B,N,R,variant) (forward calls) or
 * helio_bwd_scratch_bytes(B,N,R,variant) (backward calls) bytes the large-problem kernels first compact, per
 * image tile, the rays whose footprint is not EXACTLY zero there and walk only those.  The reference evaluates
 * every (ray, pixel) pair (:142-148); with tens of mrad of orientation error about half of the rays of a large
 * field miss the receiver by so many sigma that every one of their products underflows below half an ulp of
 * any f32 accumulator — leaving them out changes no bit of any output (doodle_amd/csrc/cull.h states the
 * criterion, which is evaluated on what the kernels compute, with a margin).  Results are IDENTICAL with and
 * without scratch, with a buffer that is too small (the dense kernels run) and with HELIO_CULL=0; a query
 * returns 0 when the kernel the call would run takes no scratch OR would not gain from it (in the backward a list
 * means fewer workgroups, which pays only where the dense grid is more than one round of the chip) — a call that
 * is handed enough scratch anyway uses it.
 */
long helio_fwd_scratch_bytes(int B, int N, int R, int variant);
long helio_bwd_scratch_bytes(int B, int N, int R, int variant);
/*
 * The part of helio_fwd_scratch_bytes a forward call cannot do without (0 for most sizes): forward variants
 * 14..17 — the 256x256 LDS-table kernel with the heliostat sum split into 2, 4, 8, 16 parts across workgroups,
 * for few images of many heliostats — keep their partial images in the scratch and return HELIO_E_SCRATCH when
 * handed less.  A caller that always passes helio_fwd_scratch_bytes() never sees that code.
 */
long helio_fwd_scratch_required(int B, int N, int R, int variant);

int         helio_abi_version(void);
const char *helio_last_error_string(void);
/* gfx arch name of device `device` into buf; HELIO_E_NODEVICE if there is none. */
int         helio_device_arch(int device, char *buf, int buflen);

/*
 * Replaces the trig of rotate_normals_batch, :87-91: for M = B*N error pairs (mrad),
 *   trig_d[m,:] = (cos(e0*1e-3), sin(e0*1e-3), cos(e1*1e-3), sin(e1*1e-3))
 * with the precise device sinf/cosf.  The table is an INPUT of the geometry kernels so that a
 * caller who needs the reference's CPU bits (parity runs) can compute it with CPU torch.
 */
int helio_error_trig(long M, const float *errs_d, float *trig_d, void *stream);

/*
 * Replaces the per-ray part of HelioField.render, :356-389 plus the per-ray
 * constants of gaussian_blur_batch :126-127,146: orientation-error rotation
 * (:78-104), leaky-ReLU Z clamp and renormalisation (:369-373), incident
 * direction (:376-380), reflection (:46-50, :383), ray/plane intersection
 * (:52-75).  Bit-identical with the reference's CPU fp32 results for `actual`
 * and `refl`.
 *
 *   helios_d [N,3]   sun_d [B,3]   action_d [B,N,3]
 *   trig_d   [B,N,4] = (cos_e, sin_e, cos_u, sin_u) of (error_mrad * 1e-3);
 *            trig_b_stride = N*4 normally, 0 to broadcast one [N,4] table
 *   actual_d [B,N,3] out            refl_d [B,N,3] out, may be NULL
 *   rays_d   [B,N,HELIO_RAY_STRIDE] out, may be NULL
 */
int helio_geometry_fwd(int B, int N,
                       const float *helios_d, const float *sun_d, const float *action_d,
                       const float *trig_d, long trig_b_stride,
                       const helio_plane *plane,
                       float *actual_d, float *refl_d, float *rays_d,
                       void *stream);

/*
 * Replaces gaussian_blur_batch + the sum over heliostats, :107-149, :404-406.
 *   image_d[b,i,j] = sum_n exp2(-((xs[i]+a)^2 + (ys[j]+b)^2 + c2) * k2)
 * xs_d/ys_d [R] are the reference's torch.linspace pixel coordinates (:129-130);
 * image dim0 runs along plane_u, dim1 along plane_v.
 * variant: 0 or 2 = f32 MFMA (kernel chosen by problem size), 1 = VALU LDS-tiled;
 * 3..6 force one MFMA kernel (regs 128x128, LDS-tile 128x128, LDS-tile 256x256, regs 64x64);
 * 9 = the k-split block kernel (one 32x32 block per workgroup, its 4 / 8 / 16 waves split the heliostat
 * sum and add their partial blocks in wave order): what 0 chooses for few images of many heliostats
 * (B*(R/32)^2 <= 1024 blocks and N >= 128, or <= 2048 blocks and N >= 500 — one sun over a whole plant;
 * HELIO_KSPLIT=0 switches it off);
 * 7, 8 = the split-bf16 kernels (opt-in, never chosen by 0/2): every f32 factor split exactly
 * into three bf16 pieces, six partial products per product on the bf16 matrix pipe, f32
 * accumulation; the dropped partial products are below 2^-23 of each product.  7 sums in two
 * levels (16 rays on the pipe, then a round-to-nearest vector add): against fp64 at N = 2000 its
 * worst per-pixel relative error is 8.6e-7, tighter than the exact-f32 MFMA kernel's one-level
 * chain (1.2e-6), at 1.77x its speed; 8 sums in one level: 2.5e-6, 2.03x.
 * 14, 15, 16, 17 = the 256x256 LDS-table kernel with the heliostat sum split into 2, 4, 8, 16 parts of consecutive
 * rays across workgroups (each part summed from zero into a partial image in the caller's scratch, the partial
 * images added in part order: bits a function of N, R and the variant): what 0 chooses for tens of images of a
 * large field (24 <= B*ceil(R/256)^2 < 192 tiles and >= 448 rays per part; HELIO_SPLIT=0 switches the choice
 * off).  These need helio_fwd_scratch_required() bytes of scratch: HELIO_E_SCRATCH otherwise.
 * helio_render_fwd / helio_env_step_fwd also take 10, 11, 12 (the single-launch block kernel with 1, 2, 4
 * waves per 32x32 block; needs N <= 64, 128, 256) and 13 (the few-ray streaming kernel; needs N <= 8,
 * R % 4 == 0 and 16-byte aligned ys / images): forced forms of what 0 chooses by size, for parity
 * tests and tuning; HELIO_E_INVALID where the form does not exist for the problem.
 */
int helio_splat_fwd(int B, int N, int R,
                    const float *rays_d, const float *xs_d, const float *ys_d,
                    float *image_d, int variant, void *scratch_d, long scratch_bytes, void *stream);

/*
 * The whole forward of HelioField.render (:356-406) in one call: helio_geometry_fwd followed
 * by helio_splat_fwd on the same stream, or — for small problems, which are launch-latency
 * bound — ONE fused kernel in which every workgroup traces the rays of its own sun.  Results
 * are identical either way.  rays_d may be NULL only if the caller never runs the backward AND
 * the problem takes the fused path; pass the work buffer to be safe.  refl_d may be NULL.
 */
int helio_render_fwd(int B, int N, int R,
                     const float *helios_d, const float *sun_d, const float *action_d,
                     const float *trig_d, long trig_b_stride, const helio_plane *plane,
                     const float *xs_d, const float *ys_d,
                     float *actual_d, float *refl_d, float *rays_d, float *image_d,
                     int variant, void *scratch_d, long scratch_bytes, void *stream);

/* Kernel launches helio_render_fwd(variant 0) issues for this size: 1 (fused) or 2. */
int helio_render_fwd_launches(int B, int N, int R);

/*
 * The variant that helio_render_fwd's variant 0 resolves to at (B, N, R): 10..13 (a form of the
 * single-launch kernel) or 3, 5, 6, 9, 14..16 (geometry + that splat kernel); 0 for invalid sizes.  Every
 * kernel sums an image's heliostats in an order that depends on N and R only, so a caller that
 * renders a batch in pieces (one shard of the sun batch per GPU) passes the choice
 * of the WHOLE batch with every piece and gets the rows of the unsharded render bit for bit.
 */
int helio_render_fwd_choice(int B, int N, int R);

/*
 * The same for the backward: the variant that helio_render_bwd's variant 0 resolves to at (B, N, R) — 8 (the
 * single-launch form), 4, 2, 12, or 9 / 10 / 11 (the small-tile kernel with its contracted axis whole / cut between
 * four / eight waves); 0 for invalid sizes.  A ray's moments are sums over ITS image only, in an order fixed by the
 * variant, N and R, so a shard that passes the whole batch's choice gets the unsharded gradient's rows bit for
 * bit (the lists of helio_bwd_scratch_bytes and the workgroup shapes still follow the shard's own size: they
 * never change a sum).  Additive in ABI version 2.
 */
int helio_render_bwd_choice(int B, int N, int R);

/* Column blocks the backward splat splits an R-wide image into (the size of the
 * second dimension of moments_d). */
int helio_splat_bwd_blocks(int R);

/*
 * Backward of helio_splat_fwd (autograd of :107-149,:404-406 w.r.t. the ray
 * parameters): per ray and per column block jb, the five centred moments of
 * Gg = grad_image[b] * gauss_bn,
 *   moments_d[b,jb,n,:] = sum_{i, j in block} Gg * (1, t, s, t^2, s^2),
 *   t = xs[i]+a, s = ys[j]+b.
 * moments_d has shape [B, helio_splat_bwd_blocks(R), N, HELIO_MOMENT_STRIDE]; a block is
 * 64 image columns for (M0, Ms, Mss) and — in the MFMA kernels — 64 image rows for (Mt, Mtt);
 * only the sum over blocks is meaningful.
 * variant: 0 = by problem size, 1 = VALU kernel, 2 = f32 MFMA kernels (256-wide tiles, two
 * launches), 3 = f32 MFMA small-tile kernel (both passes in one launch), 4 = streaming VALU kernel
 * for a handful of rays per image (bound by reading grad_image once), 5 = the split-bf16 MFMA
 * kernels (opt-in, never chosen by 0: grad-image values and factors split exactly into three
 * bf16 pieces, six partial products per product on the bf16 matrix pipe, f32 accumulation);
 * 6 / 7 = the small-tile kernel forced to 4 / 8 waves per workgroup and one ray block per wave (tests, tuning);
 * 9 / 10 / 11 = the forms 3 chooses between by size: contracted axis whole / cut between four / eight waves
 * (helio_render_bwd_choice); 12 = the LDS-tile kernels of 2 in 64-ray tiles (same bits as 2, lists as 2's; what 0 chooses
 * for fields of 33..192 heliostats once images and batch give it a few hundred workgroups, and for larger fields where
 * 256-ray tiles would pad the field or leave the chip partly idle).
 */
int helio_splat_bwd(int B, int N, int R,
                    const float *rays_d, const float *xs_d, const float *ys_d,
                    const float *grad_image_d, float *moments_d, int variant,
                    void *scratch_d, long scratch_bytes, void *stream);

/*
 * Backward of helio_geometry_fwd: chains d(image)/d(ray parameters) (from the
 * moments; moments_d may be NULL when the image has no gradient) and the direct
 * cotangents of `actual` and `refl` (either may be NULL) back to the mirror
 * normals.  Same branch semantics as torch autograd of :356-389 (leaky-ReLU
 * slope, clamp_min and where() masks).
 *   grad_action_d [B,N,3] out
 */
int helio_geometry_bwd(int B, int N, int n_blocks,
                       const float *helios_d, const float *sun_d, const float *action_d,
                       const float *trig_d, long trig_b_stride,
                       const helio_plane *plane,
                       const float *moments_d,
                       const float *grad_actual_d, const float *grad_refl_d,
                       float *grad_action_d, void *stream);

/*
 * The whole backward of HelioField.render in one call: helio_splat_bwd (skipped when
 * grad_image_d is NULL) followed by helio_geometry_bwd on the same stream.  moments_d is the
 * [B, helio_splat_bwd_blocks(R), N, 5] work buffer; any cotangent may be NULL.
 * variant: helio_splat_bwd's, plus 8 = moments AND geometry adjoint in ONE launch (a workgroup owns
 * 32 rays of a sun completely, the moments stay in LDS; R <= 256) — what 0 chooses for small,
 * latency-bound problems (B * ceil(N/32) <= 64 workgroups in the small-tile kernel's regime;
 * HELIO_BWD_FUSED=0 switches the choice off).  moments_d is not written in that form.
 * helio_env_step_bwd takes the same variants.
 */
int helio_render_bwd(int B, int N, int R,
                     const float *helios_d, const float *sun_d, const float *action_d,
                     const float *trig_d, long trig_b_stride, const helio_plane *plane,
                     const float *rays_d, const float *xs_d, const float *ys_d,
                     const float *grad_image_d, const float *grad_actual_d, const float *grad_refl_d,
                     float *moments_d, float *grad_action_d, int variant,
                     void *scratch_d, long scratch_bytes, void *stream);

/*
 * Replaces calculate_ideal_normals, :256-278:
 *   out[b,n,:] = unit( unit(sun_b - h_n) + unit(target - h_n) ), bit-identical
 *   with the reference's CPU fp32 result.
 */
int helio_ideal_normals(int B, int N, const float *helios_d, const float *sun_d,
                        const float target_position[3], float *out_d, void *stream);

/*
 * Replaces the arithmetic of init_actions, :291-304 (the random draw stays with the caller:
 * `noise_d` is the reference's torch.randn_like(ideal), M = B*N rows of 3):
 *   out[m,:] = unit( ideal[m,:] + noise[m,:] * noise_scale ),  unit(v) = v / max(|v|, 1e-9),
 * every operation rounded as the reference's CPU fp32 ops round (multiply, add, norm =
 * sqrt(fma(z,z,fma(y,y,x*x))), IEEE division): bit-identical with the reference on the same draw.
 * `out_d` may alias `ideal_d` or `noise_d`.
 */
int helio_init_actions(long M, const float *ideal_d, const float *noise_d, float noise_scale,
                       float *out_d, void *stream);

/*
 * ---- HelioEnv.step loss block (SURVEY.md §8 f) -------------------------------------------
 * Replaces test_environment.py :436-457 (peak-normalised MSE, EDT-weighted distance loss,
 * per-image mean error), :101-130 + :460-488 (boundary loss with the axes step() passes:
 * east (1,0,0), up (0,0,1)) and :132-155 + :450-455 (alignment loss).  error_mask_ratio < 0:
 * the use_error_mask=False branch; otherwise (:444-452, B <= 4096) only images whose mean error
 * exceeds torch.quantile(mean errors, 1 - ratio) enter mse and dist.
 * Two launches forward, one backward; fixed-order reductions.
 *
 *   img_d, target_d, dmaps_d [B,R,R]   tx_d [B] = clamp_min(amax(target[b]), 1e-6)  (:436)
 *   ideal_d, actual_d, action_d [B,N,3]   helios_d [N,3]
 *   workspace_d: helio_step_losses_workspace(B,N,R) floats
 *   out_d [5] = (mse, dist, bound, alignment_loss, flag) with flag = 1 if mse, dist or bound
 *               is NaN/Inf (the asserts of :495-501)
 *   mae_d [B] (monitor 'mae_image'), keep_d [B] (the 0/1 error mask; all ones without it),
 *   align_err_d [B,N] (mrad), all_bounds_d [B,N]
 *   aux_d [B, 3+3N], optional (NULL to skip): the observation row cat(sun_b, action_b) of :424,
 *   written from sun_d [B,3] and action_d by the same launch
 */
long helio_step_losses_workspace(int B, int N, int R);

/*
 * Replaces make_distance_maps, test_environment.py:92-97 (host round trip through scipy's
 * distance_transform_edt): out_d[b,i,j] = Euclidean pixel distance to the nearest pixel of
 * image b above thr·max(image b); exact (integer squared distances, fp64 sqrt).
 * workspace_d: helio_distance_maps_workspace(B,R) 4-byte words.
 */
long helio_distance_maps_workspace(int B, int R);
int helio_distance_maps(int B, int R, const float *img_d, float thr, void *workspace_d, float *out_d, void *stream);
int helio_step_losses_fwd(int B, int N, int R,
                          const float *img_d, const float *target_d, const float *tx_d, const float *dmaps_d,
                          const float *ideal_d, const float *actual_d, const float *action_d,
                          const float *helios_d, const float target_position[3], const float target_normal[3],
                          float width, float height, int exponential_risk, float error_mask_ratio,
                          float *workspace_d, float *out_d, float *mae_d, float *keep_d, float *align_err_d,
                          float *all_bounds_d, const float *sun_d, float *aux_d, void *stream);
/*
 * Backward: g_*_d are DEVICE scalars (cotangents of the four losses; NULL = none); keep_d is
 * the mask the forward wrote (NULL = all ones);
 * grad_img_d [B,R,R], grad_actual_d [B,N,3] (through the alignment loss), grad_action_d
 * [B,N,3] (through the boundary loss); any output may be NULL.
 */
int helio_step_losses_bwd(int B, int N, int R,
                          const float *img_d, const float *target_d, const float *tx_d, const float *dmaps_d,
                          const float *ideal_d, const float *actual_d, const float *action_d,
                          const float *helios_d, const float target_position[3], const float target_normal[3],
                          float width, float height, int exponential_risk,
                          const float *g_mse_d, const float *g_dist_d, const float *g_bound_d, const float *g_align_d,
                          const float *keep_d,
                          float *grad_img_d, float *grad_actual_d, float *grad_action_d, void *stream);

/*
 * HelioEnv.step forward in one call (test_environment.py:416-457): helio_render_fwd followed by
 * helio_step_losses_fwd on the image and normals it produced, arguments as in those two.  For
 * the problem sizes helio_render_fwd runs as one launch, the loss partial sums are taken in
 * that same launch from the image tile in registers (2 launches for the whole step, the image
 * is written once and never re-read); otherwise it is exactly the two calls (4 launches).
 * workspace_d: helio_env_step_workspace(B,N,R) floats.  helio_env_step_launches → 2 or 4.
 *
 * The reference asserts after every step that mse, dist and bound are finite (:495-501): six
 * device→host reads.  out_d[4] is that test as one flag; with notify != NULL (a record from
 * helio_notify_create) and ticket != 0 the finishing workgroup also publishes (flag, ticket) to
 * slot ticket % HELIO_NOTIFY_SLOTS of the record, which lives in coherent pinned host memory:
 * helio_notify_wait(record, ticket, timeout) polls it and returns the flag (0/1) without a
 * hipMemcpy or a stream synchronise (≈12 µs on an idle MI355X).  HELIO_E_STALE: more than
 * HELIO_NOTIFY_SLOTS later tickets were issued before this wait — read out_d[4] instead.
 */
#define HELIO_NOTIFY_SLOTS 64
int helio_notify_create(int **record);      /* host pointer, valid on every device */
int helio_notify_destroy(int *record);
int helio_notify_wait(const int *record, int ticket, double timeout_seconds);
long helio_env_step_workspace(int B, int N, int R);
int helio_env_step_launches(int B, int N, int R);
int helio_env_step_fwd(int B, int N, int R,
                       const float *helios_d, const float *sun_d, const float *action_d,
                       const float *trig_d, long trig_b_stride, const helio_plane *plane,
                       const float *xs_d, const float *ys_d,
                       float *actual_d, float *refl_d, float *rays_d, float *image_d, int variant,
                       const float *target_d, const float *tx_d, const float *dmaps_d, const float *ideal_d,
                       const float target_position[3], const float target_normal[3],
                       float width, float height, int exponential_risk, float error_mask_ratio,
                       float *workspace_d, float *out_d, float *mae_d, float *keep_d, float *align_err_d,
                       float *all_bounds_d, float *aux_d, int *notify, int ticket,
                       void *scratch_d, long scratch_bytes, void *stream);

/*
 * Backward of helio_env_step_fwd w.r.t. the action in one call: the cotangents of the four
 * scalars (device scalars, NULL = none; keep_d = the mask the forward wrote, NULL = all ones)
 * plus optional cotangents of `actual` and `refl` → grad_action_d [B,N,3].
 *   - through mse/dist: d/d image is formed by the loss block's adjoint and contracted with the
 *     footprints (helio_splat_bwd).  helio_env_step_bwd_image_ws(B,N,R) == 0: few rays per image,
 *     the [B,R,R] image cotangent is never materialised (it is evaluated on the fly while
 *     image / target / distance map stream through the moment kernel once) and grad_image_ws_d
 *     may be NULL; otherwise it is a [B,R,R] work buffer.
 *   - through alignment_loss and bound: the per-ray adjoints are evaluated inside the geometry
 *     backward (no extra launch, no [B,N,3] temporaries).
 * moments_d: [B, helio_splat_bwd_blocks(R), N, 5] work buffer, needed with g_mse_d or g_dist_d.
 * Launches: 1 (ray losses only), 2 (few rays) or 3.
 */
int helio_env_step_bwd_image_ws(int B, int N, int R);     /* 1: grad_image_ws_d is needed */
int helio_env_step_bwd(int B, int N, int R,
                       const float *helios_d, const float *sun_d, const float *action_d,
                       const float *trig_d, long trig_b_stride, const helio_plane *plane,
                       const float *rays_d, const float *xs_d, const float *ys_d,
                       const float *image_d, const float *target_d, const float *tx_d, const float *dmaps_d,
                       const float *ideal_d, const float target_position[3], const float target_normal[3],
                       float width, float height, int exponential_risk,
                       const float *g_mse_d, const float *g_dist_d, const float *g_bound_d, const float *g_align_d,
                       const float *keep_d, const float *grad_actual_d, const float *grad_refl_d,
                       float *grad_image_ws_d, float *moments_d, float *grad_action_d, int variant,
                       void *scratch_d, long scratch_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HELIO_H */
