"""ctypes binding of libhelio.so (include/helio.h) for torch tensors.

This is the only place the package touches the C ABI.  It hands the library raw device
pointers of caller-owned torch tensors plus torch's current HIP stream; it never
falls back to a CPU implementation: if the library or a HIP device is missing, every
op raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhelio.so")

RAY_STRIDE = 4
MOMENT_STRIDE = 5

EXPORTS = (
    "helio_abi_version", "helio_last_error_string", "helio_device_arch", "helio_error_trig", "helio_geometry_fwd",
    "helio_splat_fwd", "helio_render_fwd", "helio_render_bwd", "helio_render_fwd_launches", "helio_splat_bwd_blocks", "helio_splat_bwd", "helio_geometry_bwd",
    "helio_ideal_normals", "helio_init_actions", "helio_step_losses_workspace", "helio_step_losses_fwd", "helio_step_losses_bwd",
    "helio_distance_maps_workspace", "helio_distance_maps",
    "helio_env_step_workspace", "helio_env_step_launches", "helio_render_fwd_choice", "helio_render_bwd_choice", "helio_env_step_fwd",
    "helio_notify_create", "helio_notify_destroy", "helio_notify_wait",
    "helio_env_step_bwd_image_ws", "helio_env_step_bwd", "helio_fwd_scratch_bytes", "helio_bwd_scratch_bytes",
    "helio_fwd_scratch_required",
)
ABI_VERSION = 2


class Plane(ctypes.Structure):
    """``struct helio_plane`` (host memory)."""
    _fields_ = [("origin", ctypes.c_float * 3), ("normal", ctypes.c_float * 3),
                ("u", ctypes.c_float * 3), ("v", ctypes.c_float * 3), ("w", ctypes.c_float * 3),
                ("sigma_scale", ctypes.c_float)]


_vp, _i, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
_f, _f3 = ctypes.c_float, ctypes.c_float * 3
_lib = None


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen libhelio.so and declare the prototypes of include/helio.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m doodle_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = ctypes.CDLL(path)
    pp = ctypes.POINTER(Plane)
    protos = {
        "helio_abi_version": (_i, []),
        "helio_last_error_string": (ctypes.c_char_p, []),
        "helio_device_arch": (_i, [_i, ctypes.c_char_p, _i]),
        "helio_error_trig": (_i, [_l, _vp, _vp, _vp]),
        "helio_geometry_fwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _l, pp, _vp, _vp, _vp, _vp]),
        "helio_splat_fwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _l, _vp]),
        "helio_render_fwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _l, pp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _l, _vp]),
        "helio_render_bwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _l, pp] + [_vp] * 8 + [_i, _vp, _l, _vp]),
        "helio_fwd_scratch_bytes": (_l, [_i, _i, _i, _i]),
        "helio_bwd_scratch_bytes": (_l, [_i, _i, _i, _i]),
        "helio_fwd_scratch_required": (_l, [_i, _i, _i, _i]),
        "helio_render_fwd_launches": (_i, [_i, _i, _i]),
        "helio_splat_bwd_blocks": (_i, [_i]),
        "helio_splat_bwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _l, _vp]),
        "helio_geometry_bwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _l, pp, _vp, _vp, _vp, _vp, _vp]),
        "helio_ideal_normals": (_i, [_i, _i, _vp, _vp, ctypes.c_float * 3, _vp, _vp]),
        "helio_init_actions": (_i, [_l, _vp, _vp, _f, _vp, _vp]),
        "helio_distance_maps_workspace": (_l, [_i, _i]),
        "helio_distance_maps": (_i, [_i, _i, _vp, _f, _vp, _vp, _vp]),
        "helio_step_losses_workspace": (_l, [_i, _i, _i]),
        "helio_step_losses_fwd": (_i, [_i, _i, _i] + [_vp] * 8 + [_f3, _f3, _f, _f, _i, _f] + [_vp] * 8 + [_vp]),
        "helio_step_losses_bwd": (_i, [_i, _i, _i] + [_vp] * 8 + [_f3, _f3, _f, _f, _i] + [_vp] * 5 + [_vp] * 3 + [_vp]),
        "helio_env_step_workspace": (_l, [_i, _i, _i]),
        "helio_env_step_launches": (_i, [_i, _i, _i]),
        "helio_render_fwd_choice": (_i, [_i, _i, _i]),
        "helio_render_bwd_choice": (_i, [_i, _i, _i]),
        "helio_env_step_fwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _l, pp, _vp, _vp, _vp, _vp, _vp, _vp, _i]
                               + [_vp] * 4 + [_f3, _f3, _f, _f, _i, _f] + [_vp] * 7 + [_vp, _i, _vp, _l, _vp]),
        "helio_env_step_bwd_image_ws": (_i, [_i, _i, _i]),
        "helio_env_step_bwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _l, pp] + [_vp] * 8 + [_f3, _f3, _f, _f, _i]
                               + [_vp] * 10 + [_i, _vp, _l, _vp]),
        "helio_notify_create": (_i, [ctypes.POINTER(_vp)]),
        "helio_notify_destroy": (_i, [_vp]),
        "helio_notify_wait": (_i, [_vp, _i, ctypes.c_double]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.helio_abi_version() != ABI_VERSION:
        raise RuntimeError("libhelio.so ABI version mismatch")
    _lib = lib
    return lib


def _check(lib, rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libhelio: {lib.helio_last_error_string().decode()} (code {rc})")


def _dev(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise RuntimeError("doodle_amd renders only on a HIP device (MI355X); got a CPU tensor — "
                           "there is no CPU fallback")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """torch's current HIP stream handle (the raw getter is ~30x cheaper than building a
    torch.cuda.Stream object per call, which matters for the launch-bound small configs)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _load_hostbind():
    """The compiled binding of the same C ABI (csrc/hostbind.cpp), if it was built.  It only
    trims host time (≈5 µs per call); HELIO_HOSTBIND=0 forces the ctypes path."""
    if os.environ.get("HELIO_HOSTBIND", "1") == "0":
        return None
    try:
        from . import _hostbind
        return _hostbind if _hostbind.abi_version() == ABI_VERSION else None
    except ImportError:
        return None


def _plane_handle(hb, plane: "Plane") -> int:
    h = getattr(plane, "_hb_handle", None)
    if h is None:
        h = hb.make_plane(list(plane.origin) + list(plane.normal) + list(plane.u) + list(plane.v) + list(plane.w)
                          + [plane.sigma_scale])
        plane._hb_handle = h
    return h


class HipOps:
    """Tensor-level front end of the C ABI.  All tensors fp32, contiguous, on one HIP device."""

    def __setattr__(self, name, value):
        # compiled render contexts bake in the forced kernel variant and belong to one binding: reassigning
        # either retires every context built so far (csrc/hostbind.cpp, g_generation)
        if name in ("splat_variant", "bwd_variant", "hb"):
            for hb in (self.__dict__.get("hb"), value if name == "hb" else None):
                if hb is not None and hasattr(hb, "invalidate_contexts"):
                    hb.invalidate_contexts()
        object.__setattr__(self, name, value)
        if name in ("cull", "hb"):
            hb = self.__dict__.get("hb")
            if hb is not None and hasattr(hb, "set_use_scratch"):
                hb.set_use_scratch(bool(self.__dict__.get("cull", True)))

    def __init__(self):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: doodle_amd has no CPU fallback")
        buf = ctypes.create_string_buffer(64)
        _check(self.lib, self.lib.helio_device_arch(torch.cuda.current_device(), buf, 64))
        self.arch = buf.value.decode()
        if not self.arch.startswith("gfx950"):
            raise RuntimeError(f"libhelio.so is built for gfx950 (MI355X); device is {self.arch}")
        # hand the footprint kernels a device scratch buffer, with which the large-problem kernels skip the rays
        # that are exactly zero on a tile (include/helio.h "Device scratch"; identical results either way);
        # False: every call runs dense — A/B runs and the dense roofline measurement (HELIO_CULL=0 does the
        # same inside the library)
        self.cull = True
        self.splat_variant = int(os.environ.get("HELIO_SPLAT_VARIANT", "0"))
        self.bwd_variant = int(os.environ.get("HELIO_BWD_VARIANT", "0"))
        self.hb = _load_hostbind()
        self._notify, self._ticket = None, 0      # completion record (helio_notify_create), lazily

    def error_trig(self, errs):
        """[..., 2] mrad error angles on the device → [..., 4] (cos_e, sin_e, cos_u, sin_u)."""
        errs = errs.detach().to(torch.float32).contiguous()
        trig = torch.empty(errs.shape[:-1] + (4,), dtype=torch.float32, device=errs.device)
        _check(self.lib, self.lib.helio_error_trig(errs.numel() // 2, _dev(errs), trig.data_ptr(), _stream()))
        return trig

    # -- forward ---------------------------------------------------------------------------
    def geometry_fwd(self, helios, sun, normals, trig, trig_b_stride, plane, want_refl=True, want_rays=True):
        B, N = normals.shape[0], normals.shape[1]
        actual = torch.empty_like(normals)
        refl = torch.empty_like(normals) if want_refl else None
        rays = torch.empty((B, N, RAY_STRIDE), dtype=torch.float32, device=normals.device) if want_rays else None
        _check(self.lib, self.lib.helio_geometry_fwd(
            B, N, _dev(helios), _dev(sun), _dev(normals), _dev(trig), trig_b_stride, plane,
            actual.data_ptr(), refl.data_ptr() if want_refl else None,
            rays.data_ptr() if want_rays else None, _stream()))
        return actual, refl, rays

    def _scratch(self, query, B, N, R, variant, like, cull=None):
        """→ (tensor keeping it alive | None, pointer | None, bytes) of the optional device scratch."""
        if self.cull if cull is None else cull:
            n = query(int(B), int(N), int(R), int(variant))
        else:       # dense: only what the kernel cannot do without (the partial images of a split heliostat sum)
            n = self.lib.helio_fwd_scratch_required(int(B), int(N), int(R), int(variant)) if query is self.lib.helio_fwd_scratch_bytes else 0
        if n <= 0:
            return None, None, 0
        t = torch.empty(n, dtype=torch.uint8, device=like.device)
        return t, t.data_ptr(), n

    def render_fwd(self, helios, sun, normals, trig, trig_b_stride, plane, xs, ys, want_refl=True, rays=None,
                   variant=None):
        """geometry + splat in ONE C call (one fused launch for small problems).
        ``rays``: a caller-provided [B,N,4] work buffer to reuse, else a fresh one.
        ``variant``: a forced kernel for this call (None: ``self.splat_variant``)."""
        variant = self.splat_variant if variant is None else int(variant)
        if self.hb is not None:
            return self.hb.render_fwd(_plane_handle(self.hb, plane), helios, sun, normals, trig, trig_b_stride,
                                      xs, ys, rays, want_refl, variant)
        B, N, R = normals.shape[0], normals.shape[1], xs.shape[0]
        actual = torch.empty_like(normals)
        refl = torch.empty_like(normals) if want_refl else None
        if rays is None:
            rays = torch.empty((B, N, RAY_STRIDE), dtype=torch.float32, device=normals.device)
        image = torch.empty((B, R, R), dtype=torch.float32, device=normals.device)
        _keep, sp, sn = self._scratch(self.lib.helio_fwd_scratch_bytes, B, N, R, variant, normals)
        _check(self.lib, self.lib.helio_render_fwd(
            B, N, R, _dev(helios), _dev(sun), _dev(normals), _dev(trig), trig_b_stride, plane,
            _dev(xs), _dev(ys), actual.data_ptr(), refl.data_ptr() if want_refl else None,
            rays.data_ptr(), image.data_ptr(), variant, sp, sn, _stream()))
        return image, actual, refl, rays

    def render_choice(self, B, N, R):
        """The variant ``helio_render_fwd``'s 0 resolves to at (B, N, R) (``helio_render_fwd_choice``): what a
        shard of a batch of B suns forces so that it reproduces the unsharded render's rows bit for bit."""
        return int(self.lib.helio_render_fwd_choice(int(B), int(N), int(R)))

    def render_bwd_choice(self, B, N, R):
        """The variant ``helio_render_bwd``'s 0 resolves to at (B, N, R) (``helio_render_bwd_choice``): a shard that
        passes the whole batch's choice gets the unsharded gradient's rows bit for bit."""
        return int(self.lib.helio_render_bwd_choice(int(B), int(N), int(R)))

    def render_context(self, field, trig, trig_b_stride):
        """A compiled render context of ``field`` for the trig table ``trig`` (csrc/hostbind.cpp
        RenderCtx): ``ctx.render(sun, action, want_refl)`` is HelioField.render's no-autograd call with
        two tensor arguments.  None when the compiled binding is not built."""
        hb = self.hb
        if hb is None:
            return None
        return hb.RenderCtx(_plane_handle(hb, field._plane), field.heliostat_positions, field._xs, field._ys, trig,
                            trig_b_stride, self.splat_variant)

    def render_nograd(self, field, sun, action, trig, trig_b_stride, want_refl):
        """Fast path of HelioField.render without autograd for torch.Tensor inputs: dtype /
        device / shape fix-ups, allocation and the launch all happen in the compiled binding.
        Returns None when that binding is not built (the caller then takes the general path)."""
        hb = self.hb
        if hb is None:
            return None
        plane = field._plane
        handle = getattr(plane, "_hb_handle", None) or _plane_handle(hb, plane)
        out = hb.render_any(handle, field.heliostat_positions, sun, action, trig, trig_b_stride, field._xs, field._ys,
                            field._ray_ws, want_refl, self.splat_variant)
        field._ray_ws = out[3]
        return out

    # -- completion record of the env step (include/helio.h, helio_notify_*) --------------------
    def _next_ticket(self):
        """→ (record pointer, ticket) for one helio_env_step_fwd call."""
        if self._notify is None:
            rec = _vp()
            _check(self.lib, self.lib.helio_notify_create(ctypes.byref(rec)))
            self._notify = rec.value
        self._ticket = self._ticket % 0x7FFFFFFF + 1          # never 0
        return self._notify, self._ticket

    def notify_wait(self, ticket, timeout=30.0):
        """The NaN/Inf flag of the env step issued with ``ticket``, polled from pinned host memory
        (no hipMemcpy, no stream synchronise).  None when the slot is no longer available or the
        step has not finished in ``timeout`` seconds: read the device flag instead."""
        if self.hb is not None:
            rc = self.hb.notify_wait(self._notify, ticket, timeout)
        else:
            rc = self.lib.helio_notify_wait(self._notify, ticket, timeout)
        return bool(rc) if rc >= 0 else None

    def env_step_fwd(self, helios, sun, normals, trig, trig_b_stride, plane, xs, ys, c, rays=None, want_aux=False,
                     notify=False):
        """HelioEnv.step forward — render + loss block (+ the `aux` observation row) in ONE C call
        (``helio_env_step_fwd``: 2 launches for small problems, the loss partials taken from the
        image tile in registers).  → (image, actual, refl, rays, out[5], mae, angles, all_bounds,
        keep, aux | None, ticket) — ``ticket`` (0 without ``notify``) is for ``notify_wait``."""
        rec, ticket = self._next_ticket() if notify else (0, 0)
        if self.hb is not None:
            return (*self.hb.env_step_core(_plane_handle(self.hb, plane), helios, sun, normals, trig, trig_b_stride,
                                           xs, ys, rays, self.splat_variant, c.target, c.tx, c.dmaps, c.ideal,
                                           c.tp_l, c.tn_l, c.W, c.H, c.exp_risk, c.mask_ratio,
                                           bool(want_aux), rec, ticket), ticket)
        B, N, R = normals.shape[0], normals.shape[1], xs.shape[0]
        dev = normals.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        actual, refl, image = torch.empty_like(normals), torch.empty_like(normals), new(B, R, R)
        if rays is None:
            rays = new(B, N, RAY_STRIDE)
        ws = new(self.lib.helio_env_step_workspace(B, N, R))
        out, mae, keep, align, allb = new(5), new(B), new(B), new(B, N), new(B, N)
        aux = new(B, 3 + 3 * N) if want_aux else None
        _keep, sp, sn = self._scratch(self.lib.helio_fwd_scratch_bytes, B, N, R, self.splat_variant, normals)
        _check(self.lib, self.lib.helio_env_step_fwd(
            B, N, R, _dev(helios), _dev(sun), _dev(normals), _dev(trig), trig_b_stride, plane, _dev(xs), _dev(ys),
            actual.data_ptr(), refl.data_ptr(), rays.data_ptr(), image.data_ptr(), self.splat_variant,
            _dev(c.target), _dev(c.tx), _dev(c.dmaps), _dev(c.ideal), c.tp, c.tn, c.W, c.H, int(c.exp_risk),
            c.mask_ratio, ws.data_ptr(), out.data_ptr(), mae.data_ptr(), keep.data_ptr(), align.data_ptr(),
            allb.data_ptr(), aux.data_ptr() if want_aux else None, rec or None, ticket, sp, sn, _stream()))
        return image, actual, refl, rays, out, mae, align, allb, keep, aux, ticket

    def env_step_bwd(self, helios, sun, normals, trig, trig_b_stride, plane, rays, xs, ys, image, c, g_mse, g_dist,
                     g_bound, g_align, keep, g_actual, g_refl):
        """Backward of ``env_step_fwd`` w.r.t. the action in ONE C call (``helio_env_step_bwd``);
        every cotangent may be None.  → grad_action [B,N,3]."""
        if self.hb is not None:
            return self.hb.env_step_bwd(_plane_handle(self.hb, plane), helios, sun, normals, trig, trig_b_stride, rays,
                                        xs, ys, image, c.target, c.tx, c.dmaps, c.ideal, c.tp_l, c.tn_l, c.W,
                                        c.H, c.exp_risk, g_mse, g_dist, g_bound, g_align, keep, g_actual, g_refl,
                                        self.bwd_variant)
        B, N, R = normals.shape[0], normals.shape[1], xs.shape[0]
        grad = torch.empty_like(normals)
        moments = gws = None
        if g_mse is not None or g_dist is not None:
            moments = torch.empty((B, self.lib.helio_splat_bwd_blocks(R), N, MOMENT_STRIDE), dtype=torch.float32,
                                  device=normals.device)
            if self.lib.helio_env_step_bwd_image_ws(B, N, R) or self.bwd_variant not in (0, 4):
                gws = torch.empty_like(image)
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        _keep, sp, sn = (self._scratch(self.lib.helio_bwd_scratch_bytes, B, N, R, self.bwd_variant, normals)
                         if moments is not None else (None, None, 0))
        _check(self.lib, self.lib.helio_env_step_bwd(
            B, N, R, _dev(helios), _dev(sun), _dev(normals), _dev(trig), trig_b_stride, plane, _dev(rays), _dev(xs),
            _dev(ys), _dev(image), _dev(c.target), _dev(c.tx), _dev(c.dmaps), _dev(c.ideal), c.tp, c.tn, c.W, c.H,
            int(c.exp_risk), ptr(g_mse), ptr(g_dist), ptr(g_bound), ptr(g_align), ptr(keep), ptr(g_actual),
            ptr(g_refl), ptr(gws), ptr(moments), grad.data_ptr(), self.bwd_variant, sp, sn, _stream()))
        return grad

    # -- the two autograd nodes as C++ torch::autograd::Function (compiled binding only) ------------
    def render_node(self, field, sun, normals, trig, trig_b_stride, variant=None, bwd_variant=None):
        """``_Render.apply`` of field.py without the Python Function (None: binding not built).
        ``variant`` / ``bwd_variant``: forced forward / backward kernels for this call (None: ``self.splat_variant``
        / ``self.bwd_variant``)."""
        if self.hb is None:
            return None
        return self.hb.render_autograd(_plane_handle(self.hb, field._plane), field.heliostat_positions, sun, normals,
                                       trig, trig_b_stride, field._xs, field._ys,
                                       self.splat_variant if variant is None else int(variant),
                                       self.bwd_variant if bwd_variant is None else int(bwd_variant))

    def env_step_node(self, field, sun, normals, trig, trig_b_stride, c, notify=False):
        """``_EnvStep.apply`` of losses.py without the Python Function (None: binding not built).
        → (image, actual, refl, mse, dist, bound, alignment_loss, mae, angles, all_bounds, flag, ticket)."""
        if self.hb is None:
            return None
        rec, ticket = self._next_ticket() if notify else (0, 0)
        out = self.hb.env_step_autograd(_plane_handle(self.hb, field._plane), field.heliostat_positions, sun, normals,
                                        trig, trig_b_stride, field._xs, field._ys, self.splat_variant, self.bwd_variant,
                                        c.target, c.tx, c.dmaps, c.ideal, c.tp_l, c.tn_l, c.W, c.H,
                                        c.exp_risk, c.mask_ratio, rec, ticket)
        return (*out, ticket)

    def env_step_context(self, field, trig, trig_b_stride, c):
        """A compiled step context (csrc/hostbind.cpp EnvStepCtx) for ``field``, the trig table ``trig`` and
        the loss constants ``c``: ``ctx.step(sun, action, ticket)`` is HelioEnv.step's no-autograd call with
        three arguments (ticket 0: no completion record).  None when the compiled binding is not built."""
        hb = self.hb
        if hb is None:
            return None
        if self._notify is None:
            self._next_ticket()
        return hb.EnvStepCtx(_plane_handle(hb, field._plane), field.heliostat_positions, field._xs, field._ys, trig,
                             trig_b_stride, self.splat_variant, c.target, c.tx, c.dmaps, c.ideal, c.tp_l, c.tn_l, c.W, c.H,
                             c.exp_risk, c.mask_ratio, self._notify)

    def next_ticket(self):
        self._ticket = self._ticket % 0x7FFFFFFF + 1
        return self._ticket

    def env_step_nograd(self, field, sun, action, trig, trig_b_stride, c, notify=False):
        """HelioEnv.step without autograd in one call of the compiled binding (render + loss block +
        aux, outputs already in the shapes step() returns).  None when that binding is not built.
        → (image, actual, refl [B·N,3], mse, dist, bound, alignment_loss, flag, mae [B,1],
        angles [B·N], all_bounds [B,N], aux, normals [B,N,3], ticket)."""
        if self.hb is None:
            return None
        rec, ticket = self._next_ticket() if notify else (0, 0)
        r = self.hb.env_step_fwd(_plane_handle(self.hb, field._plane), field.heliostat_positions, sun, action, trig,
                                 trig_b_stride, field._xs, field._ys, field._ray_ws, self.splat_variant, c.target,
                                 c.tx, c.dmaps, c.ideal, c.tp_l, c.tn_l, c.W, c.H, c.exp_risk,
                                 c.mask_ratio, rec, ticket)
        field._ray_ws = r[3]
        return (*r[:3], *r[4:], ticket)

    def splat_fwd(self, rays, xs, ys, variant=None, cull=None):
        """``cull``: hand over the device scratch (None: ``self.cull``) — False runs the dense kernel."""
        B, N, R = rays.shape[0], rays.shape[1], xs.shape[0]
        variant = self.splat_variant if variant is None else variant
        image = torch.empty((B, R, R), dtype=torch.float32, device=rays.device)
        _keep, sp, sn = self._scratch(self.lib.helio_fwd_scratch_bytes, B, N, R, variant, rays, cull)
        _check(self.lib, self.lib.helio_splat_fwd(
            B, N, R, _dev(rays), _dev(xs), _dev(ys), image.data_ptr(), variant, sp, sn, _stream()))
        return image

    # -- backward --------------------------------------------------------------------------
    def splat_bwd(self, rays, xs, ys, grad_image, variant=None, cull=None):
        B, N, R = rays.shape[0], rays.shape[1], xs.shape[0]
        variant = self.bwd_variant if variant is None else variant
        jb = self.lib.helio_splat_bwd_blocks(R)
        moments = torch.empty((B, jb, N, MOMENT_STRIDE), dtype=torch.float32, device=rays.device)
        _keep, sp, sn = self._scratch(self.lib.helio_bwd_scratch_bytes, B, N, R, variant, rays, cull)
        _check(self.lib, self.lib.helio_splat_bwd(
            B, N, R, _dev(rays), _dev(xs), _dev(ys), _dev(grad_image), moments.data_ptr(), variant, sp, sn,
            _stream()))
        return moments

    def render_bwd(self, helios, sun, normals, trig, trig_b_stride, plane, rays, xs, ys, grad_image, grad_actual,
                   grad_refl, variant=None):
        """splat backward + geometry backward in ONE C call; any cotangent may be None.
        ``variant``: a forced backward kernel for this call (None: ``self.bwd_variant``)."""
        variant = self.bwd_variant if variant is None else int(variant)
        if self.hb is not None:
            return self.hb.render_bwd(_plane_handle(self.hb, plane), helios, sun, normals, trig, trig_b_stride, rays,
                                      xs, ys, grad_image, grad_actual, grad_refl, variant)
        B, N, R = normals.shape[0], normals.shape[1], xs.shape[0]
        grad = torch.empty_like(normals)
        moments = None
        if grad_image is not None:
            moments = torch.empty((B, self.lib.helio_splat_bwd_blocks(R), N, MOMENT_STRIDE), dtype=torch.float32,
                                  device=normals.device)
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        _keep, sp, sn = (self._scratch(self.lib.helio_bwd_scratch_bytes, B, N, R, variant, normals)
                         if moments is not None else (None, None, 0))
        _check(self.lib, self.lib.helio_render_bwd(
            B, N, R, _dev(helios), _dev(sun), _dev(normals), _dev(trig), trig_b_stride, plane, _dev(rays),
            _dev(xs), _dev(ys), ptr(grad_image), ptr(grad_actual), ptr(grad_refl), ptr(moments), grad.data_ptr(),
            variant, sp, sn, _stream()))
        return grad

    def geometry_bwd(self, helios, sun, normals, trig, trig_b_stride, plane, moments, grad_actual, grad_refl):
        B, N = normals.shape[0], normals.shape[1]
        grad = torch.empty_like(normals)
        _check(self.lib, self.lib.helio_geometry_bwd(
            B, N, moments.shape[1] if moments is not None else 0,
            _dev(helios), _dev(sun), _dev(normals), _dev(trig), trig_b_stride, plane,
            _dev(moments) if moments is not None else None,
            _dev(grad_actual) if grad_actual is not None else None,
            _dev(grad_refl) if grad_refl is not None else None,
            grad.data_ptr(), _stream()))
        return grad

    # -- ideal normals ---------------------------------------------------------------------
    def ideal_normals(self, helios, sun, target_xyz):
        if self.hb is not None:
            return self.hb.ideal_normals(helios, sun, list(target_xyz))
        B, N = sun.shape[0], helios.shape[0]
        out = torch.empty((B, N, 3), dtype=torch.float32, device=helios.device)
        _check(self.lib, self.lib.helio_ideal_normals(
            B, N, _dev(helios), _dev(sun), (ctypes.c_float * 3)(*target_xyz), out.data_ptr(), _stream()))
        return out

    def init_actions(self, ideal, noise, scale):
        """unit(ideal + noise·scale) row by row with the reference's roundings (helio_init_actions)."""
        ideal, noise = ideal.contiguous(), noise.contiguous()
        out = torch.empty_like(ideal)
        _check(self.lib, self.lib.helio_init_actions(ideal.numel() // 3, _dev(ideal), _dev(noise), float(scale),
                                                     out.data_ptr(), _stream()))
        return out

    # -- distance maps -----------------------------------------------------------------------
    def distance_maps(self, imgs, thr=0.5):
        B, R = imgs.shape[0], imgs.shape[-1]
        imgs = imgs.contiguous()
        ws = torch.empty(self.lib.helio_distance_maps_workspace(B, R), dtype=torch.int32, device=imgs.device)
        out = torch.empty_like(imgs)
        _check(self.lib, self.lib.helio_distance_maps(B, R, _dev(imgs), thr, ws.data_ptr(), out.data_ptr(), _stream()))
        return out

    # -- HelioEnv.step loss block --------------------------------------------------------------
    def step_losses_fwd(self, img, actual, action, c):
        """``c``: the env's per-sun constants (target, tx, dmaps, ideal, helios, tp, tn, W, H, exp_risk,
        mask_ratio).  Returns out[5] = (mse, dist, bound, alignment_loss, nonfinite flag), mae [B],
        alignment errors [B,N] (mrad), boundary terms [B,N], keep [B] (the 0/1 error mask)."""
        if self.hb is not None:
            return self.hb.step_losses_fwd(img, c.target, c.tx, c.dmaps, c.ideal, actual, action, c.helios,
                                           c.tp_l, c.tn_l, c.W, c.H, c.exp_risk, c.mask_ratio)
        B, N, R = action.shape[0], action.shape[1], img.shape[-1]
        dev = img.device
        ws = torch.empty(self.lib.helio_step_losses_workspace(B, N, R), dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        mae = torch.empty(B, dtype=torch.float32, device=dev)
        keep = torch.empty(B, dtype=torch.float32, device=dev)
        align = torch.empty((B, N), dtype=torch.float32, device=dev)
        allb = torch.empty((B, N), dtype=torch.float32, device=dev)
        _check(self.lib, self.lib.helio_step_losses_fwd(
            B, N, R, _dev(img), _dev(c.target), _dev(c.tx), _dev(c.dmaps), _dev(c.ideal), _dev(actual),
            _dev(action), _dev(c.helios), c.tp, c.tn, c.W, c.H, int(c.exp_risk), c.mask_ratio,
            ws.data_ptr(), out.data_ptr(), mae.data_ptr(), keep.data_ptr(), align.data_ptr(), allb.data_ptr(),
            None, None, _stream()))
        return out, mae, align, allb, keep

    def step_losses_bwd(self, img, actual, action, c, g_mse, g_dist, g_bound, g_align, keep, want_img, want_actual,
                        want_action):
        if self.hb is not None:
            return self.hb.step_losses_bwd(img, c.target, c.tx, c.dmaps, c.ideal, actual, action, c.helios,
                                           c.tp_l, c.tn_l, c.W, c.H, c.exp_risk, g_mse, g_dist,
                                           g_bound, g_align, keep, bool(want_img), bool(want_actual),
                                           bool(want_action))
        B, N, R = action.shape[0], action.shape[1], img.shape[-1]
        grad_img = torch.empty_like(img) if want_img else None
        grad_actual = torch.empty_like(actual) if want_actual else None
        grad_action = torch.empty_like(action) if want_action else None
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        _check(self.lib, self.lib.helio_step_losses_bwd(
            B, N, R, _dev(img), _dev(c.target), _dev(c.tx), _dev(c.dmaps), _dev(c.ideal), _dev(actual),
            _dev(action), _dev(c.helios), c.tp, c.tn, c.W, c.H, int(c.exp_risk),
            ptr(g_mse), ptr(g_dist), ptr(g_bound), ptr(g_align), ptr(keep), ptr(grad_img), ptr(grad_actual),
            ptr(grad_action), _stream()))
        return grad_img, grad_actual, grad_action


_ops = None


def get_ops() -> HipOps:
    """The process-wide HipOps (raises if the library or the device is missing)."""
    global _ops
    if _ops is None:
        _ops = HipOps()
    return _ops
