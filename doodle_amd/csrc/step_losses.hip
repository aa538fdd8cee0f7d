// Fused loss block of HelioEnv.step for gfx950 (SURVEY.md §8 f, rows 1, 2 and 4).
//
// Reference: test_environment.py :436-457 (peak-normalised MSE, |pred-targ|, per-image mean,
// EDT-weighted distance loss), :101-130 (boundary / anti-spill loss), :132-155 (alignment
// angle).  The reference runs this as ≈60 elementwise/reduction launches over [B,R,R] and
// [B,N] tensors; here it is two launches forward (partials, then one finishing workgroup) and
// one launch backward.  The image part is genuinely HBM-bound: it streams img, target and the
// distance map once (12 B/pixel, float4 loads) and writes nothing but partial sums.
// Reductions are two-stage in a fixed order (no atomics): bit-reproducible run to run.
#include <hip/hip_runtime.h>
#include "helio.h"
#include "step_loss_math.h"

namespace helio {

constexpr int SL_THREADS = 256;
constexpr int SL_PIX_PER_WG = 4096;     // 16 pixels per thread
constexpr int SL_RAYS_PER_WG = 256;

// grid.x = image workgroups (B·chunks) followed by ray workgroups
__global__ void __launch_bounds__(SL_THREADS)
step_losses_partial(int B, int N, int R, int chunks, const float* __restrict__ img,
                    const float* __restrict__ target, const float* __restrict__ tx,
                    const float* __restrict__ dmaps, const float* __restrict__ ideal,
                    const float* __restrict__ actual, const float* __restrict__ action,
                    const float* __restrict__ helios, LossGeom g,
                    float* __restrict__ part_img,      // [B, chunks, 3]
                    float* __restrict__ part_ray,      // [ray_wgs, 2]
                    float* __restrict__ align_err, float* __restrict__ all_bounds,
                    const float* __restrict__ sun, float* __restrict__ aux) {
    __shared__ float scratch[4];
    const int img_wgs = B * chunks;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < img_wgs) {
        const int b = blockIdx.x / chunks, ch = blockIdx.x % chunks;
        const long P = (long)R * R;
        const long base = (long)b * P;
        const float s = tx[b];
        float sq = 0.f, ab = 0.f, ds = 0.f;
        auto one = [&](float x, float y, float dm) {
            const float d = x / s - y / s;       // pred_n - targ_n, as the reference divides (:438-441)
            const float e = fabsf(d);
            sq = __builtin_fmaf(d, d, sq);
            ab += e;
            ds = __builtin_fmaf(e, dm, ds);
        };
        const long p0 = (long)ch * SL_PIX_PER_WG;
        if ((P & 3) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long p = p0 + 4l * (tid + SL_THREADS * k);
                if (p < P) {
                    const float4 a = *reinterpret_cast<const float4*>(img + base + p);
                    const float4 t = *reinterpret_cast<const float4*>(target + base + p);
                    const float4 m = *reinterpret_cast<const float4*>(dmaps + base + p);
                    one(a.x, t.x, m.x); one(a.y, t.y, m.y); one(a.z, t.z, m.z); one(a.w, t.w, m.w);
                }
            }
        } else {
            for (int k = 0; k < 16; ++k) {
                const long p = p0 + tid + (long)SL_THREADS * k;
                if (p < P) one(img[base + p], target[base + p], dmaps[base + p]);
            }
        }
        sq = block_sum(sq, scratch);
        ab = block_sum(ab, scratch);
        ds = block_sum(ds, scratch);
        if (tid == 0) {
            float* o = part_img + 3l * blockIdx.x;
            o[0] = sq; o[1] = ab; o[2] = ds;
        }
    } else {
        const int wg = blockIdx.x - img_wgs;
        const long m = (long)wg * SL_RAYS_PER_WG + tid;
        float sa = 0.f, sb = 0.f;
        if (m < (long)B * N) {
            const int n = (int)(m % N);
            const RayLoss r = ray_loss(ideal + 3 * m, actual + 3 * m, action + 3 * m, helios + 3l * n, g);
            align_err[m] = r.ang;
            all_bounds[m] = r.out;
            if (aux) {      // the observation's `aux` row: [sun_b, action_b] (test_environment.py:424)
                const int b = (int)(m / N);
                float* a = aux + (long)b * (3 + 3l * N);
                a[3 + 3 * n] = action[3 * m]; a[4 + 3 * n] = action[3 * m + 1]; a[5 + 3 * n] = action[3 * m + 2];
                if (n == 0) { a[0] = sun[3 * b]; a[1] = sun[3 * b + 1]; a[2] = sun[3 * b + 2]; }
            }
            sa = r.ang;
            sb = g.exponential_risk ? expf(r.out + 1e-6f) : r.out;
        }
        sa = block_sum(sa, scratch);
        sb = block_sum(sb, scratch);
        if (tid == 0) { part_ray[2l * wg] = sa; part_ray[2l * wg + 1] = sb; }
    }
}

// one workgroup: out[0..3] = mse, dist, bound, alignment_loss; out[4] = 1 if any of the first
// three is NaN/Inf else 0 (the reference's six asserts, :495-501, in one flag); mae[b];
// keep[b] = 1, or — use_error_mask, :444-447 — 1 only for the images whose mean error exceeds
// torch.quantile(mae, 1 - ratio) (linear interpolation between order statistics; B <= 4096)
constexpr int SL_MAX_MASK_B = 4096;

__global__ void __launch_bounds__(SL_THREADS)
step_losses_final(int B, int N, int R, int chunks, int ray_wgs, float mask_ratio, const float* __restrict__ part_img,
                  const float* __restrict__ part_ray, float* __restrict__ out, float* __restrict__ mae,
                  float* __restrict__ keep, int* notify, int ticket) {
    __shared__ double red[4][SL_THREADS / 64];
    __shared__ float smae[SL_MAX_MASK_B];
    __shared__ float order[2];
    const int tid = threadIdx.x;
    const double P = (double)R * R;
    const bool masked = mask_ratio >= 0.0f;
    double sq = 0, ds = 0, sa = 0, sb = 0;
    // one pass over the image partials: per-image mean error, and (no mask) the two image sums
    for (int b = tid; b < B; b += SL_THREADS) {
        double bsq = 0, bab = 0, bds = 0;
        for (int c = 0; c < chunks; ++c) {
            const float* p = part_img + 3l * ((long)b * chunks + c);
            bsq += p[0]; bab += p[1]; bds += p[2];
        }
        const float m = (float)(bab / P);
        mae[b] = m;
        if (masked) smae[b] = m;
        else { keep[b] = 1.0f; sq += bsq; ds += bds; }
    }
    for (int w = tid; w < ray_wgs; w += SL_THREADS) { sa += part_ray[2l * w]; sb += part_ray[2l * w + 1]; }
    if (masked) {
        // torch.quantile(mae, q): rank = q (B-1); lerp between the two neighbouring order statistics
        const float pos = (1.0f - mask_ratio) * (float)(B - 1);
        const int lo = (int)floorf(pos), hi = (int)ceilf(pos);
        __syncthreads();
        for (int b = tid; b < B; b += SL_THREADS) {
            const float v = smae[b];
            int rank = 0;
            for (int j = 0; j < B; ++j) rank += (smae[j] < v) || (smae[j] == v && j < b);
            if (rank == lo) order[0] = v;
            if (rank == hi) order[1] = v;
        }
        __syncthreads();
        const float w = pos - (float)lo;
        const float cutoff = w < 0.5f ? order[0] + w * (order[1] - order[0]) : order[1] - (order[1] - order[0]) * (1.0f - w);
        for (int b = tid; b < B; b += SL_THREADS) {
            const float k = smae[b] > cutoff ? 1.0f : 0.0f;
            keep[b] = k;
            if (k != 0.0f)
                for (int c = 0; c < chunks; ++c) {
                    const float* p = part_img + 3l * ((long)b * chunks + c);
                    sq += p[0]; ds += p[2];
                }
        }
    }
    // fixed-order fp64 reduction: lanes by shuffles, then the four waves
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sq += __shfl_xor(sq, d); ds += __shfl_xor(ds, d); sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = sq; red[1][tid >> 6] = ds; red[2][tid >> 6] = sa; red[3][tid >> 6] = sb; }
    __syncthreads();
    if (tid == 0) {
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
        const double M = (double)B * N;
        const float mse = (float)(t[0] / (B * P)), dist = (float)(t[1] / B);
        const float bound = (float)(t[3] / M), align = (float)(t[2] / M);
        out[0] = mse; out[1] = dist; out[2] = bound; out[3] = align;
        const bool bad = !(isfinite(mse) && isfinite(dist) && isfinite(bound));
        out[4] = bad ? 1.0f : 0.0f;
        if (notify) {       // coherent pinned host memory: (flag, ticket), ticket released last
            int* slot = notify + 2 * (ticket & (HELIO_NOTIFY_SLOTS - 1));
            slot[0] = bad ? 1 : 0;
            __hip_atomic_store(slot + 1, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// backward: cotangents of the four scalars (device pointers, may be null) → grad_img,
// grad_actual (alignment), grad_action (boundary).  Same grid split as the forward.
__global__ void __launch_bounds__(SL_THREADS)
step_losses_bwd(int B, int N, int R, int chunks, const float* __restrict__ img,
                const float* __restrict__ target, const float* __restrict__ tx,
                const float* __restrict__ dmaps, const float* __restrict__ ideal,
                const float* __restrict__ actual, const float* __restrict__ action,
                const float* __restrict__ helios, LossGeom g,
                const float* __restrict__ g_mse, const float* __restrict__ g_dist,
                const float* __restrict__ g_bound, const float* __restrict__ g_align,
                const float* __restrict__ keep,
                float* __restrict__ grad_img, float* __restrict__ grad_actual, float* __restrict__ grad_action) {
    const int img_wgs = grad_img ? B * chunks : 0;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < img_wgs) {
        const int b = blockIdx.x / chunks, ch = blockIdx.x % chunks;
        const long P = (long)R * R, base = (long)b * P;
        float s, km, kd;
        LossGradArgs{img, target, tx, dmaps, keep, g_mse, g_dist}.constants(b, B, P, s, km, kd);
        auto one = [&](float x, float y, float dm) -> float { return loss_grad_pixel(x, y, dm, s, km, kd); };
        const long p0 = (long)ch * SL_PIX_PER_WG;
        if ((P & 3) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long p = p0 + 4l * (tid + SL_THREADS * k);
                if (p < P) {
                    const float4 a = *reinterpret_cast<const float4*>(img + base + p);
                    const float4 t = *reinterpret_cast<const float4*>(target + base + p);
                    const float4 m = *reinterpret_cast<const float4*>(dmaps + base + p);
                    *reinterpret_cast<float4*>(grad_img + base + p) =
                        make_float4(one(a.x, t.x, m.x), one(a.y, t.y, m.y), one(a.z, t.z, m.z), one(a.w, t.w, m.w));
                }
            }
        } else {
            for (int k = 0; k < 16; ++k) {
                const long p = p0 + tid + (long)SL_THREADS * k;
                if (p < P) grad_img[base + p] = one(img[base + p], target[base + p], dmaps[base + p]);
            }
        }
    } else {
        const long m = (long)(blockIdx.x - img_wgs) * SL_RAYS_PER_WG + tid;
        if (m >= (long)B * N) return;
        const int n = (int)(m % N);
        const float* v = action + 3 * m;
        const float* id = ideal + 3 * m;
        const RayLoss r = ray_loss(id, actual + 3 * m, v, helios + 3l * n, g);
        const float inv = 1.0f / ((float)B * (float)N);
        float ga[3], gv[3];
        ray_loss_bwd(r, id, v, g, (g_align ? *g_align : 0.0f) * inv, (g_bound ? *g_bound : 0.0f) * inv, ga, gv);
        if (grad_actual) { grad_actual[3 * m] = ga[0]; grad_actual[3 * m + 1] = ga[1]; grad_actual[3 * m + 2] = ga[2]; }
        if (grad_action) { grad_action[3 * m] = gv[0]; grad_action[3 * m + 1] = gv[1]; grad_action[3 * m + 2] = gv[2]; }
    }
}

int step_losses_max_mask_batch() { return SL_MAX_MASK_B; }
int step_losses_chunks(int R) { return (int)(((long)R * R + SL_PIX_PER_WG - 1) / SL_PIX_PER_WG); }
int step_losses_ray_wgs(int B, int N) { return (int)(((long)B * N + SL_RAYS_PER_WG - 1) / SL_RAYS_PER_WG); }

void launch_step_losses_final(int B, int N, int R, int chunks, int ray_wgs, float mask_ratio, const float* part_img,
                              const float* part_ray, float* out, float* mae, float* keep, int* notify, int ticket,
                              hipStream_t st) {
    hipLaunchKernelGGL(step_losses_final, dim3(1), dim3(SL_THREADS), 0, st, B, N, R, chunks, ray_wgs, mask_ratio,
                       part_img, part_ray, out, mae, keep, notify, ticket);
}

void launch_step_losses_fwd(int B, int N, int R, const float* img, const float* target, const float* tx,
                            const float* dmaps, const float* ideal, const float* actual, const float* action,
                            const float* helios, const float* tp, const float* tn, float W, float H,
                            int exponential_risk, float mask_ratio, float* workspace, float* out, float* mae,
                            float* keep, float* align_err, float* all_bounds, const float* sun, float* aux,
                            int* notify, int ticket, hipStream_t st) {
    const int chunks = step_losses_chunks(R), rw = step_losses_ray_wgs(B, N);
    float* part_img = workspace;
    float* part_ray = workspace + 3l * B * chunks;
    hipLaunchKernelGGL(step_losses_partial, dim3(B * chunks + rw), dim3(SL_THREADS), 0, st, B, N, R, chunks, img,
                       target, tx, dmaps, ideal, actual, action, helios, make_geom(tp, tn, W, H, exponential_risk),
                       part_img, part_ray, align_err, all_bounds, sun, aux);
    launch_step_losses_final(B, N, R, chunks, rw, mask_ratio, part_img, part_ray, out, mae, keep, notify, ticket, st);
}

void launch_step_losses_bwd(int B, int N, int R, const float* img, const float* target, const float* tx,
                            const float* dmaps, const float* ideal, const float* actual, const float* action,
                            const float* helios, const float* tp, const float* tn, float W, float H,
                            int exponential_risk, const float* g_mse, const float* g_dist, const float* g_bound,
                            const float* g_align, const float* keep, float* grad_img, float* grad_actual,
                            float* grad_action, hipStream_t st) {
    const int chunks = step_losses_chunks(R), rw = step_losses_ray_wgs(B, N);
    // ray workgroups only when a ray gradient is asked for (helio_env_step_bwd folds them into the
    // geometry backward and passes neither)
    const int grid = (grad_img ? B * chunks : 0) + ((grad_actual || grad_action) ? rw : 0);
    if (grid == 0) return;
    hipLaunchKernelGGL(step_losses_bwd, dim3(grid), dim3(SL_THREADS), 0, st, B, N, R, chunks, img, target, tx, dmaps,
                       ideal, actual, action, helios, make_geom(tp, tn, W, H, exponential_risk), g_mse, g_dist,
                       g_bound, g_align, keep, grad_img, grad_actual, grad_action);
}

}  // namespace helio
