// _hostbind — a compiled host-side binding of the C ABI (include/helio.h) for torch tensors.
//
// The C ABI stays the boundary (raw pointers, sizes, hipStream_t).  doodle_amd/native.py binds
// it with ctypes, which is the reference binding; this module binds the SAME
// entry points from C++ so that the per-call host work of the launch-bound small configurations
// (output allocation, pointer extraction, stream lookup, argument marshalling) costs ≈2 µs
// instead of ≈7 µs.  No arithmetic happens here.  Optional: if it is not built, native.py's
// ctypes path is used.
#include <torch/csrc/autograd/anomaly_mode.h>
#include <torch/extension.h>
#include <hip/hip_runtime.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/python_variable.h>

#include <array>
#include <atomic>

#include <chrono>
#include <stdexcept>
#include <string>
#include <vector>

#include "helio.h"

namespace {

void check(int rc) {
    if (rc != 0) throw std::runtime_error(std::string("libhelio: ") + helio_last_error_string() + " (code " + std::to_string(rc) + ")");
}

const float* fp(const at::Tensor& t, const char* what) {
    TORCH_CHECK(t.is_cuda(), "doodle_amd renders only on a HIP device (MI355X); ", what, " is a CPU tensor — there is no CPU fallback");
    TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous(), what, " must be contiguous float32");
    return t.data_ptr<float>();
}
const float* fpo(const c10::optional<at::Tensor>& t, const char* what) { return t.has_value() ? fp(*t, what) : nullptr; }

void* cur_stream(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

// The optional device scratch of the footprint entry points (include/helio.h, "Device scratch"): a block from
// torch's caching allocator when the kernels the call will run can use one (large problems: the lists of rays
// that are not exactly zero on a tile), nothing otherwise — the query is a few integer operations.  The block
// is free again when the call returns: the allocator's stream ordering covers its reuse.
// set_use_scratch(false) makes every call run dense (A/B runs, the dense roofline measurement): only what a
// kernel cannot do without (helio_fwd_scratch_required: the partial images of a split heliostat sum) is handed over.
std::atomic<bool> g_use_scratch{true};
struct Scratch {
    at::Tensor t;
    void* p = nullptr;
    long bytes = 0;
    Scratch(long n, const at::Tensor& like, long required = 0) {
        if (!g_use_scratch.load(std::memory_order_relaxed)) n = required;
        if (n > 0) {
            t = at::empty({(int64_t)n}, like.options().dtype(at::kByte));
            p = t.data_ptr();
            bytes = n;
        }
    }
};
Scratch fwd_scratch(int64_t B, int64_t N, int64_t R, int64_t variant, const at::Tensor& like) {
    return Scratch(helio_fwd_scratch_bytes((int)B, (int)N, (int)R, (int)variant), like,
                   helio_fwd_scratch_required((int)B, (int)N, (int)R, (int)variant));
}
Scratch bwd_scratch(int64_t B, int64_t N, int64_t R, int64_t variant, const at::Tensor& like) {
    return Scratch(helio_bwd_scratch_bytes((int)B, (int)N, (int)R, (int)variant), like);
}

// helio_plane structs live for the life of the process (a field keeps its handle)
int64_t make_plane(const std::vector<double>& v) {
    TORCH_CHECK(v.size() == 16, "plane needs 16 numbers");
    auto* p = new helio_plane;
    for (int k = 0; k < 3; ++k) {
        p->origin[k] = (float)v[k]; p->normal[k] = (float)v[3 + k]; p->u[k] = (float)v[6 + k];
        p->v[k] = (float)v[9 + k]; p->w[k] = (float)v[12 + k];
    }
    p->sigma_scale = (float)v[15];
    return reinterpret_cast<int64_t>(p);
}

int64_t current_stream_handle(const at::Tensor& t) { return reinterpret_cast<int64_t>(cur_stream(t)); }

py::tuple render_fwd(int64_t plane, const at::Tensor& helios, const at::Tensor& sun, const at::Tensor& normals,
                     const at::Tensor& trig, int64_t trig_b_stride, const at::Tensor& xs, const at::Tensor& ys,
                     c10::optional<at::Tensor> rays_ws, bool want_refl, int64_t variant) {
    const int64_t B = normals.size(0), N = normals.size(1), R = xs.size(0);
    const float* pn = fp(normals, "action");
    at::Tensor actual = at::empty_like(normals);
    at::Tensor refl = want_refl ? at::empty_like(normals) : at::Tensor();
    at::Tensor rays = rays_ws.has_value() ? *rays_ws : at::empty({B, N, HELIO_RAY_STRIDE}, normals.options());
    at::Tensor image = at::empty({B, R, R}, normals.options());
    const Scratch sc = fwd_scratch(B, N, R, variant, normals);
    check(helio_render_fwd((int)B, (int)N, (int)R, fp(helios, "heliostat_positions"), fp(sun, "sun"), pn,
                           fp(trig, "trig"), (long)trig_b_stride, reinterpret_cast<const helio_plane*>(plane),
                           fp(xs, "xs"), fp(ys, "ys"), actual.data_ptr<float>(),
                           want_refl ? refl.data_ptr<float>() : nullptr, rays.data_ptr<float>(),
                           image.data_ptr<float>(), (int)variant, sc.p, sc.bytes, cur_stream(normals)));
    if (want_refl) return py::make_tuple(image, actual, refl, rays);
    return py::make_tuple(image, actual, py::none(), rays);
}

// The no-autograd render of tensors that may still need a dtype/device/shape fix-up
// (HelioField.render's as_tensor / reshape / contiguous, :326-337), done here instead of in Python.
py::tuple render_any(int64_t plane, const at::Tensor& helios, const at::Tensor& sun_in, const at::Tensor& action_in,
                     const at::Tensor& trig, int64_t trig_b_stride, const at::Tensor& xs, const at::Tensor& ys,
                     c10::optional<at::Tensor> rays_ws, bool want_refl, int64_t variant) {
    const int64_t N = helios.size(0), R = xs.size(0);
    // the common call needs no fix-up at all: float32, on the field's device, contiguous, [B,3] and
    // [B,3N] (or [B,N,3]) — the kernels only see pointers, so no reshaped view is built either
    if (sun_in.dim() == 2 && sun_in.scalar_type() == at::kFloat && action_in.scalar_type() == at::kFloat &&
        sun_in.device() == helios.device() && action_in.device() == helios.device() && sun_in.is_contiguous() &&
        action_in.is_contiguous() && action_in.numel() == sun_in.size(0) * N * 3) {
        const int64_t B = sun_in.size(0);
        const auto opt = helios.options();
        at::Tensor actual = at::empty({B, N, 3}, opt);
        at::Tensor refl = want_refl ? at::empty({B, N, 3}, opt) : at::Tensor();
        at::Tensor rays = (rays_ws.has_value() && rays_ws->size(0) == B && rays_ws->device() == helios.device())
                              ? *rays_ws : at::empty({B, N, HELIO_RAY_STRIDE}, opt);
        at::Tensor image = at::empty({B, R, R}, opt);
        const Scratch sc = fwd_scratch(B, N, R, variant, helios);
        check(helio_render_fwd((int)B, (int)N, (int)R, helios.data_ptr<float>(), sun_in.data_ptr<float>(),
                               action_in.data_ptr<float>(), fp(trig, "trig"), (long)trig_b_stride,
                               reinterpret_cast<const helio_plane*>(plane), xs.data_ptr<float>(), ys.data_ptr<float>(),
                               actual.data_ptr<float>(), want_refl ? refl.data_ptr<float>() : nullptr,
                               rays.data_ptr<float>(), image.data_ptr<float>(), (int)variant, sc.p, sc.bytes,
                               cur_stream(helios)));
        if (want_refl) return py::make_tuple(image, actual, refl, rays);
        return py::make_tuple(image, actual, py::none(), rays);
    }
    const auto opt = helios.options();
    at::Tensor sun = sun_in.to(opt, /*non_blocking=*/false, /*copy=*/false);
    if (sun.dim() == 1) sun = sun.unsqueeze(0);
    sun = sun.contiguous();
    const int64_t B = sun.size(0);
    at::Tensor normals = action_in.to(opt, false, false).reshape({B, N, 3}).contiguous();
    if (rays_ws.has_value() && (rays_ws->size(0) != B || rays_ws->device() != normals.device())) rays_ws.reset();
    return render_fwd(plane, helios, sun, normals, trig, trig_b_stride, xs, ys, rays_ws, want_refl, variant);
}

// Outputs of a launch-bound call come from torch's caching allocator through bare TensorImpls over blocks the
// binding asks the allocator for directly: an at::empty costs 0.72 µs (dispatcher, device guard, allocator,
// TensorImpl) next to a 3.3 µs launch and a 3.7 µs kernel, a direct allocator call and a TensorImpl about a
// third of that — and a SECOND allocator call per render was measured at 0.29 µs (6.59 → 6.13 M frames/s at
// config 2).  A block goes back to the allocator when the last tensor over it dies, so what shares a block
// decides what a kept tensor pins:
//   * the image always has a block of its own (it is what a rollout keeps; nothing small may pin it);
//   * everything small — actual | refl, the five 0-dim metrics, the per-image / per-ray monitor vectors with the
//     call's workspace, the `aux` row — is a SLOT of a slab: one allocator block serves the next up to 16 calls
//     (at most kSlabBytes), so the allocator is called once per 16 calls for them, and a tensor somebody keeps
//     pins at most its slab (≤ 256 KB) — `metrics['mse']` of 10 000 steps costs 10 000 x 256 bytes when all are
//     kept, never 10 000 images (tests/test_round3_gpu.py).
// A slab is used on the stream it was allocated under (a call on another stream starts a new one), so the caching
// allocator's stream bookkeeping — and record_stream() on any tensor of it — stays exact, as with at::empty.
constexpr int64_t kCarveMaxBytes = 8 << 20;
constexpr int64_t kSlabBytes = 256 << 10;
constexpr int64_t kSlabMaxSlots = 16;

struct Carver {
    c10::DispatchKeySet keys, keys_inference;   // of what at::empty returns outside / inside torch.inference_mode()
    caffe2::TypeMeta dtype;
    c10::Device device;
    c10::Allocator* alloc;         // the allocator at::empty itself uses for this device (on ROCm: the caching
                                   // allocator that labels its blocks with torch's "cuda" device type)
    explicit Carver(const at::Tensor& like) : dtype(like.dtype()), device(like.device()), alloc(nullptr) {
        {
            c10::InferenceMode off(false);
            const at::Tensor proto = at::empty({0}, like.options());
            keys = proto.key_set();
            alloc = proto.storage().allocator();
        }
        {
            c10::InferenceMode on(true);
            keys_inference = at::empty({0}, like.options()).key_set();
        }
        TORCH_CHECK(alloc != nullptr, "no allocator behind at::empty on ", device);
    }
    static int64_t pad(int64_t n) { return (n + 63) & ~int64_t(63); }            // 256-byte sections
    c10::Storage block(int64_t floats) const {
        c10::hip::OptionalHIPGuard guard;
        if (c10::hip::current_device() != device.index()) guard.set_index(device.index());
        const size_t bytes = (size_t)floats * sizeof(float);
        return c10::Storage(c10::Storage::use_byte_size_t(), bytes, alloc->allocate(bytes), alloc, /*resizable=*/false);
    }
    at::Tensor tensor(const c10::Storage& st, int64_t offset, at::IntArrayRef sizes) const {
        at::Tensor t = at::detail::make_tensor<c10::TensorImpl>(
            c10::Storage(st), c10::InferenceMode::is_enabled() ? keys_inference : keys, dtype);
        auto* impl = t.unsafeGetTensorImpl();
        impl->set_storage_offset(offset);
        impl->set_sizes_contiguous(sizes);
        return t;
    }
};

// 0 when `stream` is not being captured into a HIP graph, else the capture's id + 1: memory handed out during a
// capture must come from that graph's private pool (allocated during it) and must not be used after it
uint64_t capture_key(void* stream) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo((hipStream_t)stream, &status, &id) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return status == hipStreamCaptureStatusNone ? 0 : (uint64_t)id + 1;
}

// One slab in use per purpose and thread (never destroyed: no allocator call at thread or process exit).
struct Slab {
    c10::Storage st;
    int64_t slot_floats = 0, slots = 0, used = 0;
    void* stream = nullptr;
    uint64_t capture = 0;
    int device = -1;
    // storage and offset (in floats) of a fresh slot of `floats` floats for a call on `stream_` (capture key `cap`)
    std::pair<c10::Storage, int64_t> take(const Carver& cv, int64_t floats, void* stream_, uint64_t cap) {
        floats = Carver::pad(floats);
        if (!st || slot_floats != floats || used == slots || stream != stream_ || capture != cap || device != cv.device.index()) {
            slots = std::max<int64_t>(1, std::min<int64_t>(kSlabMaxSlots, kSlabBytes / (floats * (int64_t)sizeof(float))));
            st = cv.block(floats * slots);
            slot_floats = floats; used = 0; stream = stream_; capture = cap; device = cv.device.index();
        }
        return {st, (used++) * floats};
    }
};
enum SlabPurpose { kSlabRays = 0, kSlabScalars, kSlabVectors, kSlabAux, kSlabPurposes };
Slab& slab_for(SlabPurpose p) {
    thread_local Slab* slabs = new Slab[kSlabPurposes];
    return slabs[p];
}

// bumped whenever something a bound context was built from is reassigned on the Python side (the forced
// kernel variant, the binding itself); a context of an older generation declines its fast entry
std::atomic<int64_t> g_generation{0};

// one Carver per device, made on first use (the key set and allocator of a plain float32 tensor there)
const Carver& carver_for(const at::Tensor& like) {
    static std::array<std::atomic<const Carver*>, 64> table{};
    const int idx = like.get_device();
    TORCH_CHECK(idx >= 0 && idx < 64 && like.scalar_type() == at::kFloat, "carver_for: float32 device tensor expected");
    const Carver* c = table[idx].load(std::memory_order_acquire);
    if (c == nullptr) {
        const Carver* fresh = new Carver(like);
        if (table[idx].compare_exchange_strong(c, fresh, std::memory_order_acq_rel)) c = fresh;
        else delete fresh;
    }
    return *c;
}

// a tensor of its own over `base`'s storage, `delta` elements past base's first (base contiguous): what
// select / narrow / view return, minus the dispatcher and the view bookkeeping — for outputs of the no-grad
// paths only (nothing tracks that the two alias)
at::Tensor bare_view(const at::Tensor& base, int64_t delta, at::IntArrayRef sizes) {
    return carver_for(base).tensor(base.storage(), base.storage_offset() + delta, sizes);
}

// A field's render context: everything of HelioField.render's no-autograd call that does not change
// from call to call (plane, heliostats, pixel coordinates, the trig table of the current errors),
// bound once — the per-call binding then converts two tensor arguments instead of eleven.  At
// config 2 the GPU needs ≈3.7 µs per call, so each of those conversions is visible.
struct RenderCtx {
    int64_t plane;
    at::Tensor helios, xs, ys, trig, rays_ws;
    int64_t trig_b_stride, variant;
    const Carver& carve;           // the device's (made once per process)
    int64_t generation;
    at::Tensor errs;               // the error tensor the trig table was made from (bind_errors), for render_checked
    int64_t errs_version = -1;
    int64_t scratch_B = -1, scratch_need = 0;      // helio_fwd_scratch_bytes of the last batch size asked for
    Scratch scratch_for(int64_t B, int64_t N, int64_t R) {
        if (B != scratch_B) { scratch_B = B; scratch_need = helio_fwd_scratch_bytes((int)B, (int)N, (int)R, (int)variant); }
        if (scratch_need <= 0) return Scratch(0, helios);
        return fwd_scratch(B, N, R, variant, helios);
    }
    RenderCtx(int64_t plane_, at::Tensor helios_, at::Tensor xs_, at::Tensor ys_, at::Tensor trig_, int64_t stride_,
              int64_t variant_)
        : plane(plane_), helios(std::move(helios_)), xs(std::move(xs_)), ys(std::move(ys_)), trig(std::move(trig_)),
          trig_b_stride(stride_), variant(variant_), carve(carver_for(helios)), generation(g_generation.load()) {
        fp(helios, "heliostat_positions"); fp(xs, "xs"); fp(ys, "ys"); fp(trig, "trig");
    }
    void bind_errors(const at::Tensor& e) { errs = e; errs_version = (int64_t)e._version(); }
    // the receiver's tensors the plane record was made from (HelioField.target_position, target_normal, plane_u,
    // plane_v): render_checked declines once one of them has been written in place
    std::vector<at::Tensor> receiver;
    std::vector<int64_t> receiver_versions;
    void bind_receiver(const std::vector<at::Tensor>& ts) {
        receiver = ts;
        receiver_versions.clear();
        for (const at::Tensor& t : ts) receiver_versions.push_back((int64_t)t._version());
    }

    struct Outputs { at::Tensor image, actual, refl; };
    // image [B,R,R] ([R,R] when !batched), actual [B,N,3], refl [B·N,3] (the reference's monitor shape)
    Outputs outputs(int64_t B, int64_t N, int64_t R, bool want_refl, bool batched) const {
        Outputs o;
        const int64_t ni = B * R * R, na = B * N * 3;
        const int64_t rays_total = Carver::pad(na) + (want_refl ? na : 0);
        if ((ni + rays_total) * (int64_t)sizeof(float) <= kCarveMaxBytes) {
            // the image alone (what a caller keeps as an observation); actual | refl a slot of a slab
            const c10::Storage si = carve.block(ni);
            o.image = batched ? carve.tensor(si, 0, {B, R, R}) : carve.tensor(si, 0, {R, R});
            void* const stream = cur_stream(helios);
            const auto slot = slab_for(kSlabRays).take(carve, rays_total, stream, capture_key(stream));
            o.actual = carve.tensor(slot.first, slot.second, {B, N, 3});
            if (want_refl) o.refl = carve.tensor(slot.first, slot.second + Carver::pad(na), {B * N, 3});
        } else {
            const auto opt = helios.options();
            o.actual = at::empty({B, N, 3}, opt);
            if (want_refl) o.refl = at::empty({B * N, 3}, opt);
            o.image = batched ? at::empty({B, R, R}, opt) : at::empty({R, R}, opt);
        }
        return o;
    }
    // ONE no-autograd render path: the argument test both entry points share …
    // (plain float32 tensors on the field's device, contiguous, sun [B,3] (or [3] when !batched), B·N·3 action values,
    // and a batch size this context's trig table serves) → B, or -1: the caller takes the general path
    int64_t conforming_batch(const at::Tensor& sun, const at::Tensor& action, bool strict_trig) const {
        const int64_t N = helios.size(0);
        const bool batched = sun.dim() == 2;
        if (!(batched ? sun.size(1) == 3 : (sun.dim() == 1 && sun.size(0) == 3))) return -1;
        const int64_t B = batched ? sun.size(0) : 1;
        if (!(sun.scalar_type() == at::kFloat && action.scalar_type() == at::kFloat && sun.device() == helios.device() &&
              action.device() == helios.device() && sun.is_contiguous() && action.is_contiguous() &&
              action.numel() == B * N * 3))
            return -1;
        if (trig_b_stride != 0 && trig.numel() < B * N * 4) return -1;
        // the memoised entry also holds the table to the batch sizes the reference's rule gives it (:340-353)
        if (strict_trig && (trig_b_stride == 0 ? B != 1 : B < 2)) return -1;
        return B;
    }
    // … and the call itself: allocation, scratch, helio_render_fwd
    Outputs launch(const at::Tensor& sun, const at::Tensor& action, int64_t B, bool want_refl, bool batched) {
        const int64_t N = helios.size(0), R = xs.size(0);
        Outputs o = outputs(B, N, R, want_refl, batched);
        if (!rays_ws.defined() || rays_ws.size(0) != B) rays_ws = at::empty({B, N, HELIO_RAY_STRIDE}, helios.options());
        const Scratch sc = scratch_for(B, N, R);
        check(helio_render_fwd((int)B, (int)N, (int)R, helios.data_ptr<float>(), sun.data_ptr<float>(),
                               action.data_ptr<float>(), trig.data_ptr<float>(), (long)trig_b_stride,
                               reinterpret_cast<const helio_plane*>(plane), xs.data_ptr<float>(), ys.data_ptr<float>(),
                               o.actual.data_ptr<float>(), want_refl ? o.refl.data_ptr<float>() : nullptr,
                               rays_ws.data_ptr<float>(), o.image.data_ptr<float>(), (int)variant, sc.p, sc.bytes,
                               cur_stream(helios)));
        return o;
    }
    // HelioField.render's whole no-autograd call for the arguments as the caller passed them, in the reference's
    // return shapes: (image, actual) or (image, actual, refl [B·N,3]); a 1-D sun gives image [R,R], actual [1,N,3].
    // None — the caller then takes the general path — unless: the arguments conform (conforming_batch); no gradient
    // is being recorded for the action; neither the error tensor this context was bound to nor a tensor of the
    // receiver has been written since; nothing the context was built from has been reassigned (generation).
    py::object render_checked(py::handle sun_h, py::handle action_h, bool want_refl) {
        if (!THPVariable_CheckExact(sun_h.ptr()) || !THPVariable_CheckExact(action_h.ptr())) return py::none();
        const at::Tensor& sun = THPVariable_Unpack(sun_h.ptr());
        const at::Tensor& action = THPVariable_Unpack(action_h.ptr());
        if (generation != g_generation.load(std::memory_order_relaxed) || !errs.defined() ||
            (int64_t)errs._version() != errs_version)
            return py::none();
        for (size_t k = 0; k < receiver.size(); ++k)
            if ((int64_t)receiver[k]._version() != receiver_versions[k]) return py::none();
        if (action.requires_grad() && at::GradMode::is_enabled()) return py::none();
        const int64_t B = conforming_batch(sun, action, /*strict_trig=*/true);
        if (B < 0) return py::none();
        Outputs o = launch(sun, action, B, want_refl, sun.dim() == 2);
        if (want_refl) return py::make_tuple(o.image, o.actual, o.refl);
        return py::make_tuple(o.image, o.actual);
    }
    // → (image [B,R,R], actual [B,N,3]) or, with want_refl, (image, actual, refl [B,N,3]) for a [B,3] sun; None when
    // the tensors need a dtype / device / layout fix-up (the caller then takes the general path)
    py::object render(const at::Tensor& sun, const at::Tensor& action, bool want_refl) {
        const int64_t B = sun.dim() == 2 ? conforming_batch(sun, action, /*strict_trig=*/false) : -1;
        if (B < 0) return py::none();
        Outputs o = launch(sun, action, B, want_refl, true);
        if (want_refl) return py::make_tuple(o.image, o.actual, o.refl.view({B, helios.size(0), 3}));
        return py::make_tuple(o.image, o.actual);
    }
    // Measurement aid (tools/launch_floor.py): wall-clock nanoseconds per repetition of the pieces of
    // render() above, each looped n times on the calling thread — the two output allocations, the stream
    // lookup, and the C-ABI call with fixed output buffers.
    py::dict host_costs(const at::Tensor& sun, const at::Tensor& action, int64_t n) {
        const int64_t N = helios.size(0), R = xs.size(0), B = sun.size(0);
        const auto opt = helios.options();
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ns = [&](std::chrono::steady_clock::time_point t0) {
            return std::chrono::duration<double, std::nano>(now() - t0).count() / (double)n;
        };
        py::dict d;
        auto t0 = now();
        for (int64_t i = 0; i < n; ++i) { at::Tensor a = at::empty({B, N, 3}, opt); at::Tensor im = at::empty({B, R, R}, opt); }
        d["two_at_empty_ns"] = ns(t0);
        t0 = now();
        for (int64_t i = 0; i < n; ++i) { Outputs o = outputs(B, N, R, false, true); }
        d["carved_outputs_ns"] = ns(t0);
        t0 = now();
        void* st = nullptr;
        for (int64_t i = 0; i < n; ++i) st = cur_stream(helios);
        d["cur_stream_ns"] = ns(t0);
        at::Tensor actual = at::empty({B, N, 3}, opt), image = at::empty({B, R, R}, opt);
        if (!rays_ws.defined() || rays_ws.size(0) != B) rays_ws = at::empty({B, N, HELIO_RAY_STRIDE}, opt);
        check(hipStreamSynchronize((hipStream_t)st) == hipSuccess ? 0 : 1);
        t0 = now();
        for (int64_t i = 0; i < n; ++i)
            check(helio_render_fwd((int)B, (int)N, (int)R, helios.data_ptr<float>(), sun.data_ptr<float>(),
                                   action.data_ptr<float>(), trig.data_ptr<float>(), (long)trig_b_stride,
                                   reinterpret_cast<const helio_plane*>(plane), xs.data_ptr<float>(), ys.data_ptr<float>(),
                                   actual.data_ptr<float>(), nullptr, rays_ws.data_ptr<float>(), image.data_ptr<float>(),
                                   (int)variant, nullptr, 0, st));
        d["abi_call_enqueue_ns"] = ns(t0);
        (void)hipStreamSynchronize((hipStream_t)st);
        d["abi_call_with_drain_ns"] = ns(t0);
        bind_errors(trig);
        py::object sun_o = py::cast(sun), act_o = py::cast(action);
        t0 = now();
        for (int64_t i = 0; i < n; ++i) { py::object o = render_checked(sun_o, act_o, false); }
        d["render_checked_enqueue_ns"] = ns(t0);
        (void)hipStreamSynchronize((hipStream_t)st);
        t0 = now();
        for (int64_t i = 0; i < n; ++i) { py::object o = render(sun, action, false); }
        d["render_enqueue_ns"] = ns(t0);
        (void)hipStreamSynchronize((hipStream_t)st);
        d["render_with_drain_ns"] = ns(t0);
        return d;
    }
    // Forward AND the gradient w.r.t. the action for GIVEN cotangents of (image, actual, refl) in one binding
    // call: helio_render_fwd + helio_render_bwd back to back on the current stream, no autograd graph, the
    // ray and moment buffers are the context's own scratch.  What an optimiser that already holds dL/dimage
    // needs per step (BASELINE config 3).  → (image [B,R,R], actual [B,N,3], grad_action [B,N,3]); None when
    // a tensor needs a dtype / device / layout fix-up (the caller then takes the general path).
    at::Tensor moments_ws;
    py::object render_and_grad(const at::Tensor& sun, const at::Tensor& action, c10::optional<at::Tensor> g_image,
                               c10::optional<at::Tensor> g_actual, c10::optional<at::Tensor> g_refl, int64_t bwd_variant) {
        const int64_t N = helios.size(0), R = xs.size(0);
        if (!(sun.dim() == 2 && sun.scalar_type() == at::kFloat && action.scalar_type() == at::kFloat &&
              sun.device() == helios.device() && action.device() == helios.device() && sun.is_contiguous() &&
              action.is_contiguous() && action.numel() == sun.size(0) * N * 3))
            return py::none();
        const int64_t B = sun.size(0);
        if (trig_b_stride != 0 && trig.numel() < B * N * 4) return py::none();
        auto conforms = [&](const c10::optional<at::Tensor>& g, int64_t numel) {
            return !g.has_value() || (g->scalar_type() == at::kFloat && g->device() == helios.device() && g->is_contiguous() &&
                                      g->numel() == numel);
        };
        if (!conforms(g_image, B * R * R) || !conforms(g_actual, B * N * 3) || !conforms(g_refl, B * N * 3)) return py::none();
        const auto opt = helios.options();
        at::Tensor actual = at::empty({B, N, 3}, opt);
        if (!rays_ws.defined() || rays_ws.size(0) != B) rays_ws = at::empty({B, N, HELIO_RAY_STRIDE}, opt);
        at::Tensor image = at::empty({B, R, R}, opt);
        at::Tensor grad = at::empty({B, N, 3}, opt);
        void* st = cur_stream(helios);
        const helio_plane* pl = reinterpret_cast<const helio_plane*>(plane);
        {
            const Scratch sc = fwd_scratch(B, N, R, variant, helios);
            check(helio_render_fwd((int)B, (int)N, (int)R, helios.data_ptr<float>(), sun.data_ptr<float>(),
                                   action.data_ptr<float>(), trig.data_ptr<float>(), (long)trig_b_stride, pl,
                                   xs.data_ptr<float>(), ys.data_ptr<float>(), actual.data_ptr<float>(), nullptr,
                                   rays_ws.data_ptr<float>(), image.data_ptr<float>(), (int)variant, sc.p, sc.bytes, st));
        }
        float* mom = nullptr;
        if (g_image.has_value()) {
            const int64_t jb = helio_splat_bwd_blocks((int)R);
            if (!moments_ws.defined() || moments_ws.size(0) != B || moments_ws.size(1) != jb)
                moments_ws = at::empty({B, jb, N, HELIO_MOMENT_STRIDE}, opt);
            mom = moments_ws.data_ptr<float>();
        }
        const Scratch scb = g_image.has_value() ? bwd_scratch(B, N, R, bwd_variant, helios) : Scratch(0, helios);
        check(helio_render_bwd((int)B, (int)N, (int)R, helios.data_ptr<float>(), sun.data_ptr<float>(),
                               action.data_ptr<float>(), trig.data_ptr<float>(), (long)trig_b_stride, pl,
                               rays_ws.data_ptr<float>(), xs.data_ptr<float>(), ys.data_ptr<float>(),
                               g_image.has_value() ? g_image->data_ptr<float>() : nullptr,
                               g_actual.has_value() ? g_actual->data_ptr<float>() : nullptr,
                               g_refl.has_value() ? g_refl->data_ptr<float>() : nullptr, mom, grad.data_ptr<float>(),
                               (int)bwd_variant, scb.p, scb.bytes, st));
        return py::make_tuple(image, actual, grad);
    }
};

at::Tensor render_bwd(int64_t plane, const at::Tensor& helios, const at::Tensor& sun, const at::Tensor& normals,
                      const at::Tensor& trig, int64_t trig_b_stride, const at::Tensor& rays, const at::Tensor& xs,
                      const at::Tensor& ys, c10::optional<at::Tensor> g_image, c10::optional<at::Tensor> g_actual,
                      c10::optional<at::Tensor> g_refl, int64_t variant) {
    const int64_t B = normals.size(0), N = normals.size(1), R = xs.size(0);
    at::Tensor grad = at::empty_like(normals);
    at::Tensor moments;
    if (g_image.has_value())
        moments = at::empty({B, (int64_t)helio_splat_bwd_blocks((int)R), N, HELIO_MOMENT_STRIDE}, normals.options());
    const Scratch sc = g_image.has_value() ? bwd_scratch(B, N, R, variant, normals) : Scratch(0, normals);
    check(helio_render_bwd((int)B, (int)N, (int)R, fp(helios, "heliostat_positions"), fp(sun, "sun"),
                           fp(normals, "action"), fp(trig, "trig"), (long)trig_b_stride,
                           reinterpret_cast<const helio_plane*>(plane), fp(rays, "rays"), fp(xs, "xs"), fp(ys, "ys"),
                           fpo(g_image, "grad_image"), fpo(g_actual, "grad_actual"), fpo(g_refl, "grad_refl"),
                           g_image.has_value() ? moments.data_ptr<float>() : nullptr, grad.data_ptr<float>(),
                           (int)variant, sc.p, sc.bytes, cur_stream(normals)));
    return grad;
}

py::tuple step_losses_fwd(const at::Tensor& img, const at::Tensor& target, const at::Tensor& tx, const at::Tensor& dmaps,
                          const at::Tensor& ideal, const at::Tensor& actual, const at::Tensor& action,
                          const at::Tensor& helios, const std::vector<double>& tp, const std::vector<double>& tn,
                          double W, double H, bool exp_risk, double mask_ratio,
                          c10::optional<at::Tensor> sun = c10::nullopt, c10::optional<at::Tensor> aux = c10::nullopt) {
    const int64_t B = action.size(0), N = action.size(1), R = img.size(-1);
    const float tpf[3] = {(float)tp[0], (float)tp[1], (float)tp[2]}, tnf[3] = {(float)tn[0], (float)tn[1], (float)tn[2]};
    auto opt = img.options();
    at::Tensor ws = at::empty({helio_step_losses_workspace((int)B, (int)N, (int)R)}, opt);
    at::Tensor out = at::empty({5}, opt), mae = at::empty({B}, opt), keep = at::empty({B}, opt);
    at::Tensor align = at::empty({B, N}, opt), allb = at::empty({B, N}, opt);
    check(helio_step_losses_fwd((int)B, (int)N, (int)R, fp(img, "img"), fp(target, "target"), fp(tx, "tx"),
                                fp(dmaps, "distance_maps"), fp(ideal, "ideal"), fp(actual, "actual"),
                                fp(action, "action"), fp(helios, "heliostat_positions"), tpf, tnf, (float)W, (float)H,
                                exp_risk ? 1 : 0, (float)mask_ratio, ws.data_ptr<float>(), out.data_ptr<float>(),
                                mae.data_ptr<float>(), keep.data_ptr<float>(), align.data_ptr<float>(),
                                allb.data_ptr<float>(), fpo(sun, "sun"),
                                aux.has_value() ? aux->data_ptr<float>() : nullptr, cur_stream(img)));
    return py::make_tuple(out, mae, align, allb, keep);
}

struct LossGrads { at::Tensor img, actual, action; };

LossGrads step_losses_bwd_core(const at::Tensor& img, const at::Tensor& target, const at::Tensor& tx, const at::Tensor& dmaps,
                               const at::Tensor& ideal, const at::Tensor& actual, const at::Tensor& action,
                               const at::Tensor& helios, const std::vector<double>& tp, const std::vector<double>& tn,
                               double W, double H, bool exp_risk, c10::optional<at::Tensor> g_mse,
                               c10::optional<at::Tensor> g_dist, c10::optional<at::Tensor> g_bound,
                               c10::optional<at::Tensor> g_align, c10::optional<at::Tensor> keep, bool want_img,
                               bool want_actual, bool want_action) {
    const int64_t B = action.size(0), N = action.size(1), R = img.size(-1);
    const float tpf[3] = {(float)tp[0], (float)tp[1], (float)tp[2]}, tnf[3] = {(float)tn[0], (float)tn[1], (float)tn[2]};
    LossGrads g;
    if (want_img) g.img = at::empty_like(img);
    if (want_actual) g.actual = at::empty_like(actual);
    if (want_action) g.action = at::empty_like(action);
    check(helio_step_losses_bwd((int)B, (int)N, (int)R, fp(img, "img"), fp(target, "target"), fp(tx, "tx"),
                                fp(dmaps, "distance_maps"), fp(ideal, "ideal"), fp(actual, "actual"),
                                fp(action, "action"), fp(helios, "heliostat_positions"), tpf, tnf, (float)W, (float)H,
                                exp_risk ? 1 : 0, fpo(g_mse, "g_mse"), fpo(g_dist, "g_dist"), fpo(g_bound, "g_bound"),
                                fpo(g_align, "g_align"), fpo(keep, "keep"), want_img ? g.img.data_ptr<float>() : nullptr,
                                want_actual ? g.actual.data_ptr<float>() : nullptr,
                                want_action ? g.action.data_ptr<float>() : nullptr, cur_stream(img)));
    return g;
}

py::tuple step_losses_bwd(const at::Tensor& img, const at::Tensor& target, const at::Tensor& tx, const at::Tensor& dmaps,
                          const at::Tensor& ideal, const at::Tensor& actual, const at::Tensor& action,
                          const at::Tensor& helios, const std::vector<double>& tp, const std::vector<double>& tn,
                          double W, double H, bool exp_risk, c10::optional<at::Tensor> g_mse,
                          c10::optional<at::Tensor> g_dist, c10::optional<at::Tensor> g_bound,
                          c10::optional<at::Tensor> g_align, c10::optional<at::Tensor> keep, bool want_img,
                          bool want_actual, bool want_action) {
    const LossGrads g = step_losses_bwd_core(img, target, tx, dmaps, ideal, actual, action, helios, tp, tn, W, H, exp_risk,
                                             g_mse, g_dist, g_bound, g_align, keep, want_img, want_actual, want_action);
    auto o = [](const at::Tensor& t) -> py::object { return t.defined() ? py::cast(t) : py::none(); };
    return py::make_tuple(o(g.img), o(g.actual), o(g.action));
}

// HelioEnv.step forward (render + loss block + optional `aux` row) through helio_env_step_fwd
struct StepOut { at::Tensor image, actual, refl, rays, out, mae, align, allb, keep, aux; };

StepOut step_core(int64_t plane, const at::Tensor& helios, const at::Tensor& sun, const at::Tensor& normals,
                  const at::Tensor& trig, int64_t trig_b_stride, const at::Tensor& xs, const at::Tensor& ys,
                  c10::optional<at::Tensor> rays_ws, int64_t variant, const at::Tensor& target, const at::Tensor& tx,
                  const at::Tensor& dmaps, const at::Tensor& ideal, const std::vector<double>& tp,
                  const std::vector<double>& tn, double W, double H, bool exp_risk, double mask_ratio, bool want_aux,
                  int64_t notify, int64_t ticket) {
    const int64_t B = normals.size(0), N = normals.size(1), R = xs.size(0);
    const float tpf[3] = {(float)tp[0], (float)tp[1], (float)tp[2]}, tnf[3] = {(float)tn[0], (float)tn[1], (float)tn[2]};
    const auto opt = normals.options();
    const float* pn = fp(normals, "action");
    StepOut o;
    // Blocks from the caching allocator through the Carver (an at::empty costs 0.7 µs and a view op 0.3–0.4 µs in
    // front of two 3.4 µs launches), grouped by who keeps what (see kCarveMaxBytes): the 5 scalars alone (a
    // 256-byte block: metrics are what training loops append to lists); the call's workspace with the per-image
    // and per-ray monitor vectors; and, while the step is small (the launch-bound regime), the image alone, the
    // `aux` row alone, actual | refl together.
    const int64_t nws = helio_env_step_workspace((int)B, (int)N, (int)R);
    const int64_t ni = B * R * R, na = B * N * 3, naux = want_aux ? B * (3 + 3 * N) : 0;
    auto P = Carver::pad;
    const int64_t f_mae = P(nws), f_keep = f_mae + P(B), f_align = f_keep + P(B), f_allb = f_align + P(B * N),
                  f_small = f_allb + P(B * N);
    const bool carve_all = (f_small + P(ni) + 2 * P(na) + P(naux)) * (int64_t)sizeof(float) <= kCarveMaxBytes;
    const Carver& cv = carver_for(helios);
    void* const stream = cur_stream(normals);
    const uint64_t cap = capture_key(stream);
    {
        const auto sl = slab_for(kSlabScalars).take(cv, 64, stream, cap);
        o.out = cv.tensor(sl.first, sl.second, {5});
    }
    const auto sv = slab_for(kSlabVectors).take(cv, f_small, stream, cap);
    const c10::Storage& st = sv.first;
    const int64_t v0 = sv.second;
    at::Tensor ws = cv.tensor(st, v0, {nws});
    o.mae = cv.tensor(st, v0 + f_mae, {B});
    o.keep = cv.tensor(st, v0 + f_keep, {B});
    o.align = cv.tensor(st, v0 + f_align, {B, N});
    o.allb = cv.tensor(st, v0 + f_allb, {B, N});
    if (carve_all) {
        o.image = cv.tensor(cv.block(ni), 0, {B, R, R});
        const auto sr = slab_for(kSlabRays).take(cv, 2 * P(na), stream, cap);
        o.actual = cv.tensor(sr.first, sr.second, {B, N, 3});
        o.refl = cv.tensor(sr.first, sr.second + P(na), {B, N, 3});
        if (want_aux) {
            const auto sa = slab_for(kSlabAux).take(cv, naux, stream, cap);
            o.aux = cv.tensor(sa.first, sa.second, {B, 3 + 3 * N});
        }
    } else {
        o.actual = at::empty_like(normals);
        o.refl = at::empty_like(normals);
        o.image = at::empty({B, R, R}, opt);
        if (want_aux) o.aux = at::empty({B, 3 + 3 * N}, opt);
    }
    o.rays = rays_ws.has_value() ? *rays_ws : at::empty({B, N, HELIO_RAY_STRIDE}, opt);
    const Scratch sc = fwd_scratch(B, N, R, variant, normals);
    check(helio_env_step_fwd((int)B, (int)N, (int)R, fp(helios, "heliostat_positions"), fp(sun, "sun"), pn,
                             fp(trig, "trig"), (long)trig_b_stride, reinterpret_cast<const helio_plane*>(plane),
                             fp(xs, "xs"), fp(ys, "ys"), o.actual.data_ptr<float>(), o.refl.data_ptr<float>(),
                             o.rays.data_ptr<float>(), o.image.data_ptr<float>(), (int)variant, fp(target, "target"),
                             fp(tx, "tx"), fp(dmaps, "distance_maps"), fp(ideal, "ideal"), tpf, tnf, (float)W,
                             (float)H, exp_risk ? 1 : 0, (float)mask_ratio, ws.data_ptr<float>(),
                             o.out.data_ptr<float>(), o.mae.data_ptr<float>(), o.keep.data_ptr<float>(),
                             o.align.data_ptr<float>(), o.allb.data_ptr<float>(),
                             want_aux ? o.aux.data_ptr<float>() : nullptr, reinterpret_cast<int*>(notify), (int)ticket,
                             sc.p, sc.bytes, stream));
    return o;
}

// → (image, actual, refl, rays, out[5], mae, angles, all_bounds, keep, aux|None)
py::tuple env_step_core(int64_t plane, const at::Tensor& helios, const at::Tensor& sun, const at::Tensor& normals,
                        const at::Tensor& trig, int64_t trig_b_stride, const at::Tensor& xs, const at::Tensor& ys,
                        c10::optional<at::Tensor> rays_ws, int64_t variant, const at::Tensor& target,
                        const at::Tensor& tx, const at::Tensor& dmaps, const at::Tensor& ideal,
                        const std::vector<double>& tp, const std::vector<double>& tn, double W, double H,
                        bool exp_risk, double mask_ratio, bool want_aux, int64_t notify, int64_t ticket) {
    const StepOut o = step_core(plane, helios, sun, normals, trig, trig_b_stride, xs, ys, rays_ws, variant, target, tx, dmaps,
                                ideal, tp, tn, W, H, exp_risk, mask_ratio, want_aux, notify, ticket);
    py::object aux_o = want_aux ? py::cast(o.aux) : py::none();
    return py::make_tuple(o.image, o.actual, o.refl, o.rays, o.out, o.mae, o.align, o.allb, o.keep, aux_o);
}

// HelioEnv.step without autograd in ONE binding call: the dtype/shape fix-ups of the action
// (test_environment.py:416-424) plus env_step_core.  It only removes Python between them.
py::tuple env_step_fwd(int64_t plane, const at::Tensor& helios, const at::Tensor& sun, const at::Tensor& action_in,
                       const at::Tensor& trig, int64_t trig_b_stride, const at::Tensor& xs, const at::Tensor& ys,
                       c10::optional<at::Tensor> rays_ws, int64_t variant, const at::Tensor& target,
                       const at::Tensor& tx, const at::Tensor& dmaps, const at::Tensor& ideal,
                       const std::vector<double>& tp, const std::vector<double>& tn, double W, double H,
                       bool exp_risk, double mask_ratio, int64_t notify, int64_t ticket) {
    const int64_t B = sun.size(0), N = helios.size(0);
    at::Tensor normals;
    if (action_in.scalar_type() == at::kFloat && action_in.device() == helios.device() && action_in.is_contiguous() &&
        action_in.numel() == B * N * 3)
        normals = action_in.dim() == 3 ? action_in : action_in.view({B, N, 3});      // nothing to fix up
    else
        normals = action_in.to(helios.options(), false, false).reshape({B, N, 3}).contiguous();
    if (rays_ws.has_value() && (rays_ws->size(0) != B || rays_ws->device() != normals.device())) rays_ws.reset();
    const StepOut r = step_core(plane, helios, sun, normals, trig, trig_b_stride, xs, ys, rays_ws, variant, target, tx, dmaps,
                                ideal, tp, tn, W, H, exp_risk, mask_ratio, true, notify, ticket);
    // the shapes step() hands out (:503-514), made here: a view costs ≈0.3 µs in C++, ≈1.5 µs in Python
    const std::vector<at::Tensor> o = r.out.unbind(0);
    return py::make_tuple(r.image, r.actual, r.refl.view({B * N, 3}), r.rays, o[0], o[1], o[2], o[3], o[4],
                          r.mae.view({B, 1}), r.align.view({B * N}), r.allb, r.aux, normals);
}

// HelioEnv.step's no-autograd call with everything that does not change between steps bound once
// (fields' geometry, trig table, reference image / peaks / distance maps / ideal normals, loss
// constants, completion record): per step the binding converts two tensors and a ticket.  The host
// side of a step is what bounds it at small sizes (two launches, ≈14 µs of GPU).
struct EnvStepCtx;
py::object env_step_ctx_grad(EnvStepCtx& c, const at::Tensor& sun, const at::Tensor& action_in, int64_t bwd_variant,
                             int64_t ticket);
struct EnvStepCtx {
    int64_t plane, trig_b_stride, variant, notify;
    at::Tensor helios, xs, ys, trig, target, tx, dmaps, ideal, rays_ws;
    std::vector<double> tp, tn;
    double W, H, mask_ratio;
    bool exp_risk;
    EnvStepCtx(int64_t plane_, at::Tensor helios_, at::Tensor xs_, at::Tensor ys_, at::Tensor trig_, int64_t stride_,
               int64_t variant_, at::Tensor target_, at::Tensor tx_, at::Tensor dmaps_, at::Tensor ideal_,
               std::vector<double> tp_, std::vector<double> tn_, double W_, double H_, bool exp_risk_, double mask_ratio_,
               int64_t notify_)
        : plane(plane_), trig_b_stride(stride_), variant(variant_), notify(notify_), helios(std::move(helios_)),
          xs(std::move(xs_)), ys(std::move(ys_)), trig(std::move(trig_)), target(std::move(target_)), tx(std::move(tx_)),
          dmaps(std::move(dmaps_)), ideal(std::move(ideal_)), tp(std::move(tp_)), tn(std::move(tn_)), W(W_), H(H_),
          mask_ratio(mask_ratio_), exp_risk(exp_risk_) {}
    // → the tuple of env_step_fwd (outputs already in the shapes step() returns), or None when the
    // action needs a dtype / device / layout fix-up
    py::object step(const at::Tensor& sun, const at::Tensor& action_in, int64_t ticket) {
        const int64_t B = sun.size(0), N = helios.size(0);
        if (!(action_in.scalar_type() == at::kFloat && action_in.device() == helios.device() && action_in.is_contiguous() &&
              action_in.numel() == B * N * 3 && sun.scalar_type() == at::kFloat && sun.is_contiguous() &&
              sun.device() == helios.device()))
            return py::none();
        at::Tensor normals = action_in.dim() == 3 ? action_in : action_in.view({B, N, 3});
        c10::optional<at::Tensor> ws;
        if (rays_ws.defined() && rays_ws.size(0) == B) ws = rays_ws;
        const StepOut r = step_core(plane, helios, sun, normals, trig, trig_b_stride, xs, ys, ws, variant, target, tx, dmaps,
                                    ideal, tp, tn, W, H, exp_risk, mask_ratio, true, ticket != 0 ? notify : 0, ticket);
        rays_ws = r.rays;
        return py::make_tuple(r.image, r.actual, bare_view(r.refl, 0, {B * N, 3}), bare_view(r.out, 0, {}),
                              bare_view(r.out, 1, {}), bare_view(r.out, 2, {}), bare_view(r.out, 3, {}),
                              bare_view(r.out, 4, {}), bare_view(r.mae, 0, {B, 1}), bare_view(r.align, 0, {B * N}), r.allb,
                              r.aux, normals);
    }
};

// Backward of the env step in one call (helio_env_step_bwd): cotangents of the four scalars (0-d
// device tensors or None) and of actual / refl (or None) → grad_action [B,N,3]
at::Tensor env_step_bwd(int64_t plane, const at::Tensor& helios, const at::Tensor& sun, const at::Tensor& normals,
                        const at::Tensor& trig, int64_t trig_b_stride, const at::Tensor& rays, const at::Tensor& xs,
                        const at::Tensor& ys, const at::Tensor& image, const at::Tensor& target, const at::Tensor& tx,
                        const at::Tensor& dmaps, const at::Tensor& ideal, const std::vector<double>& tp,
                        const std::vector<double>& tn, double W, double H, bool exp_risk,
                        c10::optional<at::Tensor> g_mse, c10::optional<at::Tensor> g_dist,
                        c10::optional<at::Tensor> g_bound, c10::optional<at::Tensor> g_align,
                        c10::optional<at::Tensor> keep, c10::optional<at::Tensor> g_actual,
                        c10::optional<at::Tensor> g_refl, int64_t variant) {
    const int64_t B = normals.size(0), N = normals.size(1), R = xs.size(0);
    const float tpf[3] = {(float)tp[0], (float)tp[1], (float)tp[2]}, tnf[3] = {(float)tn[0], (float)tn[1], (float)tn[2]};
    const bool through_image = g_mse.has_value() || g_dist.has_value();
    at::Tensor grad = at::empty_like(normals), moments, gws;
    if (through_image) {
        moments = at::empty({B, (int64_t)helio_splat_bwd_blocks((int)R), N, HELIO_MOMENT_STRIDE}, normals.options());
        if (helio_env_step_bwd_image_ws((int)B, (int)N, (int)R) || (variant != 0 && variant != 4)) gws = at::empty_like(image);
    }
    const Scratch sc = through_image ? bwd_scratch(B, N, R, variant, normals) : Scratch(0, normals);
    check(helio_env_step_bwd((int)B, (int)N, (int)R, fp(helios, "heliostat_positions"), fp(sun, "sun"),
                             fp(normals, "action"), fp(trig, "trig"), (long)trig_b_stride,
                             reinterpret_cast<const helio_plane*>(plane), fp(rays, "rays"), fp(xs, "xs"), fp(ys, "ys"),
                             fp(image, "img"), fp(target, "target"), fp(tx, "tx"), fp(dmaps, "distance_maps"),
                             fp(ideal, "ideal"), tpf, tnf, (float)W, (float)H, exp_risk ? 1 : 0, fpo(g_mse, "g_mse"),
                             fpo(g_dist, "g_dist"), fpo(g_bound, "g_bound"), fpo(g_align, "g_align"), fpo(keep, "keep"),
                             fpo(g_actual, "grad_actual"), fpo(g_refl, "grad_refl"),
                             gws.defined() ? gws.data_ptr<float>() : nullptr,
                             through_image ? moments.data_ptr<float>() : nullptr, grad.data_ptr<float>(), (int)variant,
                             sc.p, sc.bytes, cur_stream(normals)));
    return grad;
}

// helio_notify_wait with the GIL released: → 0/1 flag, or a negative HELIO_E_* code
int64_t notify_wait(int64_t record, int64_t ticket, double timeout_seconds) {
    py::gil_scoped_release nogil;
    return helio_notify_wait(reinterpret_cast<const int*>(record), (int)ticket, timeout_seconds);
}

// ---------------------------------------------------------------------------------------------
// The two autograd nodes of the path as C++ torch::autograd::Function: same C-ABI calls as the
// Python Functions of field.py / losses.py (which stay, for the ctypes binding), but the engine
// runs their backward without the GIL and apply() costs ≈3 µs instead of ≈15 µs — at config 2 the
// GPU work of step + backward is ≈40 µs, so that overhead is the difference that matters.
// ---------------------------------------------------------------------------------------------
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

c10::optional<at::Tensor> opt_contig(const at::Tensor& g) {
    return g.defined() ? c10::optional<at::Tensor>(g.contiguous()) : c10::nullopt;
}

class RenderFn : public torch::autograd::Function<RenderFn> {
 public:
    static variable_list forward(AutogradContext* ctx, at::Tensor normals, int64_t plane, at::Tensor helios, at::Tensor sun,
                                 at::Tensor trig, int64_t trig_b_stride, at::Tensor xs, at::Tensor ys, int64_t variant,
                                 int64_t bwd_variant) {
        const int64_t B = normals.size(0), N = normals.size(1), R = xs.size(0);
        at::Tensor actual = at::empty_like(normals), refl = at::empty_like(normals);
        at::Tensor rays = at::empty({B, N, HELIO_RAY_STRIDE}, normals.options());
        at::Tensor image = at::empty({B, R, R}, normals.options());
        const Scratch sc = fwd_scratch(B, N, R, variant, normals);
        check(helio_render_fwd((int)B, (int)N, (int)R, fp(helios, "heliostat_positions"), fp(sun, "sun"),
                               fp(normals, "action"), fp(trig, "trig"), (long)trig_b_stride,
                               reinterpret_cast<const helio_plane*>(plane), fp(xs, "xs"), fp(ys, "ys"),
                               actual.data_ptr<float>(), refl.data_ptr<float>(), rays.data_ptr<float>(),
                               image.data_ptr<float>(), (int)variant, sc.p, sc.bytes, cur_stream(normals)));
        ctx->save_for_backward({normals, sun, trig, rays, helios, xs, ys});
        ctx->saved_data["plane"] = plane;
        ctx->saved_data["stride"] = trig_b_stride;
        ctx->saved_data["bwd_variant"] = bwd_variant;
        ctx->set_materialize_grads(false);        // unused outputs arrive undefined, not as zero tensors
        return {image, actual, refl};
    }

    static variable_list backward(AutogradContext* ctx, variable_list g) {
        const auto sv = ctx->get_saved_variables();
        variable_list out(10);
        if (!g[0].defined() && !g[1].defined() && !g[2].defined()) return out;
        // a NaN / Inf in the image cotangent: the reference gives NaN for EVERY ray (0·NaN), the lists give 0 for the rays they
        // drop.  Under torch.autograd.set_detect_anomaly(True) the cotangent is checked — one reduction and
        // a wait, in the mode where the caller asked for such checks — and a non-finite one runs this backward dense
        // (field.py's _Render.backward does the same for the ctypes binding)
        const bool dense = g[0].defined() && g_use_scratch.load(std::memory_order_relaxed) && torch::autograd::AnomalyMode::is_enabled() &&
                           !at::isfinite(g[0]).all().item<bool>();
        if (dense) g_use_scratch.store(false);
        try {
            out[0] = render_bwd(ctx->saved_data["plane"].toInt(), sv[4], sv[1], sv[0], sv[2], ctx->saved_data["stride"].toInt(),
                                sv[3], sv[5], sv[6], opt_contig(g[0]), opt_contig(g[1]), opt_contig(g[2]),
                                ctx->saved_data["bwd_variant"].toInt());
        } catch (...) {
            if (dense) g_use_scratch.store(true);
            throw;
        }
        if (dense) g_use_scratch.store(true);
        return out;
    }
};

py::tuple render_autograd(int64_t plane, const at::Tensor& helios, const at::Tensor& sun, const at::Tensor& normals,
                          const at::Tensor& trig, int64_t trig_b_stride, const at::Tensor& xs, const at::Tensor& ys,
                          int64_t variant, int64_t bwd_variant) {
    variable_list o = RenderFn::apply(normals, plane, helios, sun, trig, trig_b_stride, xs, ys, variant, bwd_variant);
    return py::make_tuple(o[0], o[1], o[2]);
}

class EnvStepFn : public torch::autograd::Function<EnvStepFn> {
 public:
    static variable_list forward(AutogradContext* ctx, at::Tensor normals, int64_t plane, at::Tensor helios, at::Tensor sun,
                                 at::Tensor trig, int64_t trig_b_stride, at::Tensor xs, at::Tensor ys, int64_t variant,
                                 int64_t bwd_variant, at::Tensor target, at::Tensor tx, at::Tensor dmaps, at::Tensor ideal,
                                 std::vector<double> tp, std::vector<double> tn, double W, double H, bool exp_risk,
                                 double mask_ratio, int64_t notify, int64_t ticket, bool want_aux) {
        // `normals`: the action in ANY contiguous shape of B·N·3 floats ([B,3N] as step() receives it: no view node
        // between the action and this one in the graph); with want_aux (step()'s own call) the outputs leave in the
        // shapes step() hands out — refl [B·N,3], mae [B,1], angles [B·N] — so that no view op follows either
        const int64_t B = sun.size(0), N = helios.size(0);
        const at::Tensor n3 = normals.dim() == 3 ? normals : bare_view(normals, 0, {B, N, 3});
        const StepOut r = step_core(plane, helios, sun, n3, trig, trig_b_stride, xs, ys, c10::nullopt, variant, target, tx,
                                    dmaps, ideal, tp, tn, W, H, exp_risk, mask_ratio, want_aux, notify, ticket);
        const at::Tensor &image = r.image, &actual = r.actual, &rays = r.rays, &out = r.out, &allb = r.allb, &keep = r.keep;
        const at::Tensor refl = want_aux ? bare_view(r.refl, 0, {B * N, 3}) : r.refl;
        const at::Tensor mae = want_aux ? bare_view(r.mae, 0, {B, 1}) : r.mae;
        const at::Tensor align = want_aux ? bare_view(r.align, 0, {B * N}) : r.align;
        ctx->save_for_backward({normals, sun, trig, rays, image, actual, keep, helios, xs, ys, target, tx, dmaps, ideal});
        ctx->saved_data["plane"] = plane;
        ctx->saved_data["stride"] = trig_b_stride;
        ctx->saved_data["bwd_variant"] = bwd_variant;
        ctx->saved_data["tp"] = tp;
        ctx->saved_data["tn"] = tn;
        ctx->saved_data["W"] = W;
        ctx->saved_data["H"] = H;
        ctx->saved_data["exp_risk"] = exp_risk;
        ctx->set_materialize_grads(false);
        const at::Tensor o[5] = {bare_view(out, 0, {}), bare_view(out, 1, {}), bare_view(out, 2, {}), bare_view(out, 3, {}),
                                 bare_view(out, 4, {})};
        ctx->mark_non_differentiable({mae, align, allb, o[4]});
        // with want_aux the observation row cat(sun, action) of test_environment.py:424 is a 12th output, written by
        // the loss launch; it is differentiable in the reference (d aux / d action = identity on its last 3N columns)
        if (want_aux) return {image, actual, refl, o[0], o[1], o[2], o[3], mae, align, allb, o[4], r.aux};
        return {image, actual, refl, o[0], o[1], o[2], o[3], mae, align, allb, o[4]};
    }

    static variable_list backward(AutogradContext* ctx, variable_list g) {
        const auto sv = ctx->get_saved_variables();
        const at::Tensor &action = sv[0], &sun = sv[1], &trig = sv[2], &rays = sv[3], &image = sv[4], &actual = sv[5],
                         &keep = sv[6], &helios = sv[7], &xs = sv[8], &ys = sv[9], &target = sv[10], &tx = sv[11],
                         &dmaps = sv[12], &ideal = sv[13];
        const int64_t Bn = sun.size(0), Nn = helios.size(0);
        const at::Tensor normals = action.dim() == 3 ? action : bare_view(action, 0, {Bn, Nn, 3});
        // the cotangents of refl [B·N,3] / actual in whatever shape the forward handed them out
        auto as3 = [&](const at::Tensor& t) { return t.defined() ? c10::optional<at::Tensor>(t.contiguous().view({Bn, Nn, 3})) : c10::nullopt; };
        auto shaped = [&](at::Tensor grad) { return grad.sizes() == action.sizes() ? grad : grad.view(action.sizes()); };
        const int64_t plane = ctx->saved_data["plane"].toInt(), stride = ctx->saved_data["stride"].toInt();
        const int64_t bwd_variant = ctx->saved_data["bwd_variant"].toInt();
        const std::vector<double> tp = ctx->saved_data["tp"].toDoubleVector(), tn = ctx->saved_data["tn"].toDoubleVector();
        const double W = ctx->saved_data["W"].toDouble(), H = ctx->saved_data["H"].toDouble();
        const bool exp_risk = ctx->saved_data["exp_risk"].toBool();
        variable_list res(23);
        auto g_mse = opt_contig(g[3]), g_dist = opt_contig(g[4]), g_bound = opt_contig(g[5]), g_align = opt_contig(g[6]);
        // a cotangent of the `aux` output: its action columns pass straight through
        auto add_aux = [&](at::Tensor grad) {
            if (g.size() > 11 && g[11].defined())
                grad = grad + g[11].narrow(1, 3, g[11].size(1) - 3).reshape(grad.sizes());
            return grad;
        };
        const bool any = g_mse.has_value() || g_dist.has_value() || g_bound.has_value() || g_align.has_value() ||
                         g[0].defined() || g[1].defined() || g[2].defined();
        if (!any) {              // only `aux` (or nothing) carries a gradient
            if (g.size() > 11 && g[11].defined()) res[0] = shaped(add_aux(at::zeros_like(normals)));
            return res;
        }
        if (!g[0].defined()) {
            // the whole backward in one C call (helio_env_step_bwd)
            res[0] = shaped(add_aux(env_step_bwd(plane, helios, sun, normals, trig, stride, rays, xs, ys, image, target, tx, dmaps,
                                                 ideal, tp, tn, W, H, exp_risk, g_mse, g_dist, g_bound, g_align, keep, as3(g[1]),
                                                 as3(g[2]), bwd_variant)));
            return res;
        }
        // an external cotangent of the image as well: loss gradients first, then the render backward
        const bool need_img = g_mse.has_value() || g_dist.has_value();
        at::Tensor g_image = g[0], g_actual = g[1], gn;
        if (need_img || g_align.has_value() || g_bound.has_value()) {
            const LossGrads l = step_losses_bwd_core(image, target, tx, dmaps, ideal, actual, normals, helios, tp, tn, W, H,
                                                     exp_risk, g_mse, g_dist, g_bound, g_align, keep, need_img,
                                                     g_align.has_value(), g_bound.has_value());
            if (need_img) g_image = g_image + l.img;
            if (g_align.has_value()) g_actual = g_actual.defined() ? g_actual + l.actual : l.actual;
            gn = l.action;
        }
        at::Tensor gr = render_bwd(plane, helios, sun, normals, trig, stride, rays, xs, ys, opt_contig(g_image),
                                   as3(g_actual), as3(g[2]), bwd_variant);
        res[0] = shaped(add_aux(gn.defined() ? gr + gn : gr));
        return res;
    }
};

// → (image, actual, refl, mse, dist, bound, alignment_loss, mae, angles, all_bounds, flag)
py::tuple env_step_autograd(int64_t plane, const at::Tensor& helios, const at::Tensor& sun, const at::Tensor& normals,
                            const at::Tensor& trig, int64_t trig_b_stride, const at::Tensor& xs, const at::Tensor& ys,
                            int64_t variant, int64_t bwd_variant, const at::Tensor& target, const at::Tensor& tx,
                            const at::Tensor& dmaps, const at::Tensor& ideal, const std::vector<double>& tp,
                            const std::vector<double>& tn, double W, double H, bool exp_risk, double mask_ratio,
                            int64_t notify, int64_t ticket) {
    variable_list o = EnvStepFn::apply(normals, plane, helios, sun, trig, trig_b_stride, xs, ys, variant, bwd_variant, target,
                                       tx, dmaps, ideal, tp, tn, W, H, exp_risk, mask_ratio, notify, ticket, false);
    py::tuple t(o.size());
    for (size_t k = 0; k < o.size(); ++k) t[k] = o[k];
    return t;
}

// HelioEnv.step WITH autograd in one binding call: the step context's constants, ONE autograd node (EnvStepFn, with
// the `aux` observation row as its 12th output), and the shapes step() hands out — what env.py otherwise does in
// Python around the node (argument shaping, torch.cat for aux, three views): ≈15 µs of the ≈65 µs of
// `env.step(a); metrics['dist'].backward()` at config 3.  → the tuple of EnvStepCtx::step, or None when the action
// needs a dtype / device / layout fix-up.
py::object env_step_ctx_grad(EnvStepCtx& c, const at::Tensor& sun, const at::Tensor& action_in, int64_t bwd_variant,
                             int64_t ticket) {
    const int64_t B = sun.size(0), N = c.helios.size(0);
    if (!(action_in.scalar_type() == at::kFloat && action_in.device() == c.helios.device() && action_in.is_contiguous() &&
          action_in.numel() == B * N * 3 && sun.scalar_type() == at::kFloat && sun.is_contiguous() &&
          sun.device() == c.helios.device()))
        return py::none();
    variable_list o = EnvStepFn::apply(action_in, c.plane, c.helios, sun, c.trig, c.trig_b_stride, c.xs, c.ys, c.variant, bwd_variant,
                                       c.target, c.tx, c.dmaps, c.ideal, c.tp, c.tn, c.W, c.H, c.exp_risk, c.mask_ratio,
                                       ticket != 0 ? c.notify : 0, ticket, true);
    // monitor['normals'] = action.view(batch_size, -1, 3) (:460): the one view op left, off the path of any loss
    at::Tensor normals = action_in.dim() == 3 ? action_in : action_in.view({B, N, 3});
    return py::make_tuple(o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[10], o[7], o[8], o[9], o[11], normals);
}

at::Tensor ideal_normals(const at::Tensor& helios, const at::Tensor& sun, const std::vector<double>& target) {
    const int64_t B = sun.size(0), N = helios.size(0);
    const float t[3] = {(float)target[0], (float)target[1], (float)target[2]};
    at::Tensor out = at::empty({B, N, 3}, helios.options());
    check(helio_ideal_normals((int)B, (int)N, fp(helios, "heliostat_positions"), fp(sun, "sun"), t,
                              out.data_ptr<float>(), cur_stream(helios)));
    return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled torch binding of libhelio.so's C ABI (same entry points as doodle_amd/native.py)";
    m.def("abi_version", []() { return helio_abi_version(); });
    m.def("make_plane", &make_plane);
    m.def("invalidate_contexts", []() { return ++g_generation; });
    m.def("set_use_scratch", [](bool on) { g_use_scratch.store(on); });
    m.def("current_stream_handle", &current_stream_handle);
    m.def("render_fwd", &render_fwd);
    m.def("render_any", &render_any);
    m.def("render_bwd", &render_bwd);
    py::class_<EnvStepCtx>(m, "EnvStepCtx")
        .def(py::init<int64_t, at::Tensor, at::Tensor, at::Tensor, at::Tensor, int64_t, int64_t, at::Tensor, at::Tensor,
                      at::Tensor, at::Tensor, std::vector<double>, std::vector<double>, double, double, bool, double,
                      int64_t>())
        .def("step", &EnvStepCtx::step)
        .def("step_grad", &env_step_ctx_grad);
    py::class_<RenderCtx>(m, "RenderCtx")
        .def(py::init<int64_t, at::Tensor, at::Tensor, at::Tensor, at::Tensor, int64_t, int64_t>())
        .def("render", &RenderCtx::render)
        .def("render_and_grad", &RenderCtx::render_and_grad)
        .def("host_costs", &RenderCtx::host_costs)
        .def("render_checked", &RenderCtx::render_checked)
        .def("bind_errors", &RenderCtx::bind_errors)
        .def("bind_receiver", &RenderCtx::bind_receiver)
        .def_readonly("trig", &RenderCtx::trig)
        .def_readonly("variant", &RenderCtx::variant);
    m.def("step_losses_fwd", &step_losses_fwd, py::arg("img"), py::arg("target"), py::arg("tx"), py::arg("dmaps"),
          py::arg("ideal"), py::arg("actual"), py::arg("action"), py::arg("helios"), py::arg("tp"), py::arg("tn"),
          py::arg("W"), py::arg("H"), py::arg("exp_risk"), py::arg("mask_ratio"), py::arg("sun") = py::none(),
          py::arg("aux") = py::none());
    m.def("step_losses_bwd", &step_losses_bwd);
    m.def("env_step_fwd", &env_step_fwd);
    m.def("env_step_core", &env_step_core);
    m.def("notify_wait", &notify_wait);
    m.def("env_step_bwd", &env_step_bwd);
    m.def("render_autograd", &render_autograd);
    m.def("env_step_autograd", &env_step_autograd);
    m.def("ideal_normals", &ideal_normals);
}
