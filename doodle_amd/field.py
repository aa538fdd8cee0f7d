"""HelioField — the reference's optics surface on top of the HIP kernels.

Mirrors ``HelioField`` of DOODLE's ``newenv_rl_test_multi_error.py`` (:154-415): same
constructor signature, methods, public attributes, error-selection rule and return
conventions, so that ``HelioEnv`` and the training scripts run on it unchanged.  All
arithmetic of ``render`` / ``calculate_ideal_normals`` runs in libhelio.so (hand-written
HIP for gfx950); this module only shapes arguments, keeps the pre-sampled error tensors
and wires autograd.  There is no CPU path: on a CPU device (or without the library) the
compute methods raise ``RuntimeError``.
"""
from __future__ import annotations

import os

import torch

from . import native

_TINY = 1e-9


def _get_ops():
    # indirection so the host-logic tests can substitute a checker backend
    return native.get_ops()


class _Render(torch.autograd.Function):
    """image, actual, refl = render(normals); differentiable w.r.t. ``normals`` only."""

    @staticmethod
    def forward(ctx, normals, field, sun, trig, trig_stride, variant=None, bwd_variant=None):
        image, actual, refl, rays = _get_ops().render_fwd(
            field.heliostat_positions, sun, normals, trig, trig_stride, field._plane, field._xs, field._ys,
            variant=variant)
        ctx.field, ctx.trig_stride, ctx.bwd_variant = field, trig_stride, bwd_variant
        ctx.consts = (field.heliostat_positions, field._plane, field._xs, field._ys)     # as rendered: a later assignment to the field does not reach this node
        ctx.save_for_backward(normals, sun, trig, rays)
        ctx.set_materialize_grads(False)          # unused outputs arrive as None, not as zero tensors
        return image, actual, refl

    @staticmethod
    def backward(ctx, g_image, g_actual, g_refl):
        normals, sun, trig, rays = ctx.saved_tensors
        if g_image is None and g_actual is None and g_refl is None:
            return None, None, None, None, None, None, None
        helios, plane, xs, ys = ctx.consts
        c = lambda g: g.contiguous() if g is not None else None  # noqa: E731
        kw = {} if ctx.bwd_variant is None else {"variant": ctx.bwd_variant}
        ops = _get_ops()
        # A NaN / Inf in the image cotangent: the reference's autograd gives NaN for EVERY ray (0·NaN), the lists give 0
        # for the rays they drop.  Under torch.autograd.set_detect_anomaly(True) — where the caller is
        # looking for exactly that — the cotangent is checked (one reduction and a wait) and such a backward runs dense.
        dense = (g_image is not None and getattr(ops, "cull", False) and torch.is_anomaly_enabled()
                 and not bool(torch.isfinite(g_image).all()))
        if dense:
            ops.cull = False
        try:
            g = ops.render_bwd(helios, sun, normals, trig, ctx.trig_stride, plane,
                               rays, xs, ys, c(g_image), c(g_actual), c(g_refl), **kw)
        finally:
            if dense:
                ops.cull = True
        return g, None, None, None, None, None, None


class HelioField:
    """Heliostat field with per-sun-position pre-sampled orientation errors
    (reference: newenv_rl_test_multi_error.py:154)."""

    def __init__(
        self,
        heliostat_positions,
        target_position,
        target_area: tuple,
        target_normal,
        error_scale_mrad: float = 1.0,
        sigma_scale: float = 0.01,
        initial_action_noise: float = 0.01,
        resolution: int = 100,
        device="cpu",
        max_batch_size: int = 25,
    ) -> None:
        self._rec = None              # (plane record, xs, ys): made from the receiver's attributes on first use
        self.device = torch.device(device)
        self.max_batch_size = int(max_batch_size)

        self.heliostat_positions = torch.as_tensor(
            heliostat_positions, dtype=torch.float32, device=self.device).contiguous()
        self.num_heliostats = self.heliostat_positions.shape[0]
        self.target_position = target_position
        self.target_width, self.target_height = target_area

        # The few constructor constants are computed with CPU torch ops — the same
        # ATen kernels the reference runs on CPU (:189-213) — and uploaded, so that they
        # carry the reference's bits whatever the device.
        tn = torch.as_tensor(target_normal, dtype=torch.float32).detach().cpu()
        tn = tn / tn.norm().clamp_min(_TINY)
        u = torch.tensor([1.0, 0.0, 0.0])
        if torch.allclose(tn, torch.tensor([0.0, 1.0, 0.0])):
            v = torch.tensor([0.0, 0.0, 1.0])
        else:
            v = torch.linalg.cross(tn, u)
            v = v / v.norm().clamp_min(_TINY)
        self.target_normal = tn
        self.plane_u, self.plane_v = u, v

        self.error_scale_mrad = float(error_scale_mrad)
        self.initial_action_noise = float(initial_action_noise)
        self.sigma_scale = float(sigma_scale)
        self.resolution = int(resolution)

        self._trig_cache = {}
        self._device_trig = os.environ.get("HELIO_DEVICE_TRIG", "0") == "1"
        self._ray_ws = None
        self._fast_render = None      # ops.render_context once resolved (False: compiled binding absent)
        self._render_ctx, self._ctx_key, self._ops = None, None, None
        self._fast = None             # the render context of the previous fast call (see __setattr__)
        self._receiver()              # (a constructor argument the kernels cannot stand for is refused here)
        self.reset_errors()
        self.initial_action = None

    def __setattr__(self, name, value):
        # any reassignment (errors, heliostats, the receiver's attributes → _plane / _xs / _ys, device_trig …)
        # retires the memoised render context; the slots of the caches themselves are exempt
        object.__setattr__(self, name, value)
        if name != "_fast":
            object.__setattr__(self, "_fast", None)

    # compiled contexts, cached tables and scratch are rebuilt on demand: a copy / a pickle carries none of them
    _TRANSIENT = ("_render_ctx", "_ctx_key", "_fast", "_ops", "_fast_render", "_ray_ws")

    def __getstate__(self):
        state = dict(self.__dict__)
        for name in self._TRANSIENT:
            state[name] = None
        state["_trig_cache"] = {}
        state["_rec"] = None              # (rebuilt from the attributes on first use)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    # -------------------------------------------------------------- the receiver, as live as the reference's
    # The reference reads target_position, target_normal, plane_u, plane_v, target_width, target_height, resolution and
    # sigma_scale from the instance at EVERY render (:387-401), so a script that assigns one of them between renders
    # sees it take effect.  Here they feed three native-side records — the plane (origin, normal, the orthonormal
    # frame u, v, w = u × v of the separable footprint, sigma_scale), the pixel coordinates of :129-130 and the
    # target of calculate_ideal_normals.  An assignment only stores the value and drops the records (as the reference
    # only stores it: two attributes that must change together, plane_u and plane_v, can be assigned one after the
    # other); the NEXT use rebuilds them — new objects, so every compiled context keyed on them is rebuilt too — and an
    # in-place write to one of the four tensors is caught by its version counter at that point as well.
    def _vector_attr(name, doc):        # noqa: N805 — class-body helper
        slot = "_" + name

        def get(self):
            return self.__dict__[slot]

        def set_(self, value):
            t = torch.as_tensor(value, dtype=torch.float32, device=self.device)
            if t.shape != (3,):
                raise ValueError(f"HelioField.{name} must have shape (3,), got {tuple(t.shape)}")
            self.__dict__[slot] = t
            self._rec = None              # (through __setattr__: the memoised context goes too)
        return property(get, set_, doc=doc)

    def _scalar_attr(name, cast, doc):  # noqa: N805
        slot = "_" + name

        def get(self):
            return self.__dict__[slot]

        def set_(self, value):
            self.__dict__[slot] = cast(value)
            self._rec = None
        return property(get, set_, doc=doc)

    target_position = _vector_attr("target_position", "centre of the receiver (:184-186); read at every render (:387)")
    target_normal = _vector_attr("target_normal", "receiver normal, as assigned — the render divides by its norm again (:60)")
    plane_u = _vector_attr("plane_u", "image dim-0 axis on the receiver (:206); must be orthonormal with plane_v when rendered")
    plane_v = _vector_attr("plane_v", "image dim-1 axis on the receiver (:207-213)")
    target_width = _scalar_attr("target_width", float, "receiver extent along plane_u [m] (:187)")
    target_height = _scalar_attr("target_height", float, "receiver extent along plane_v [m] (:187)")
    resolution = _scalar_attr("resolution", int, "pixels per side (:199)")
    sigma_scale = _scalar_attr("sigma_scale", float, "footprint sigma per metre of path (:198, read at :400)")
    del _vector_attr, _scalar_attr

    def _receiver_versions(self):
        d = self.__dict__
        return tuple(d[k]._version for k in ("_target_position", "_target_normal", "_plane_u", "_plane_v"))

    def _receiver(self):
        """→ (plane record, xs, ys) of the receiver's CURRENT attribute values: the records of the last call while
        nothing was assigned or written in place since, else new ones."""
        rec = self._rec
        if rec is not None and self._receiver_seen == self._receiver_versions():
            return rec
        d = self.__dict__
        tp, tn, u, v = (d[k].detach().cpu() for k in ("_target_position", "_target_normal", "_plane_u", "_plane_v"))
        # the footprint is evaluated in its separable form, |P_ij − x|² = (xs_i + a)² + (ys_j + b)² + c²,
        # which IS the reference's Σ diffs² (:141-144) exactly when (u, v) are orthonormal — as the constructor makes
        # them (:206-213); anything else is refused rather than rendered differently from the reference
        uu, vv, uv = float(u.double() @ u.double()), float(v.double() @ v.double()), float(u.double() @ v.double())
        if not (abs(uu - 1.0) <= 1e-5 and abs(vv - 1.0) <= 1e-5 and abs(uv) <= 1e-5):
            raise ValueError("HelioField: plane_u / plane_v must be orthonormal (|u|² = %.6g, |v|² = %.6g, u·v = %.3g): the HIP "
                             "footprint kernels evaluate the reference's Gaussian in its separable form" % (uu, vv, uv))
        R = self._resolution
        if R < 1:
            raise ValueError(f"HelioField.resolution must be >= 1, got {R}")
        w = torch.linalg.cross(u.double(), v.double()).float()
        vectors = (tuple(tp.tolist()), tuple(tn.tolist()), tuple(u.tolist()), tuple(v.tolist()), tuple(w.tolist()))
        plane = native.Plane(*vectors, self._sigma_scale)
        xs_key = (self._target_width, self._target_height, R, self.device)
        grid = d.get("_grid")
        if grid is None or grid[0] != xs_key:                 # (sigma_scale or a vector alone leaves the pixel grid as it is)
            grid = (xs_key, torch.linspace(-self._target_width / 2, self._target_width / 2, R).to(self.device),
                    torch.linspace(-self._target_height / 2, self._target_height / 2, R).to(self.device))
            self._grid = grid
        self._target_xyz = vectors[0]
        self._receiver_seen = self._receiver_versions()
        self._rec = rec = (plane, grid[1], grid[2])
        return rec

    def _override_record(self, i, value) -> None:
        rec = list(self._receiver())
        rec[i] = value
        self._rec = tuple(rec)            # (stays until the next assignment to a receiver attribute)

    # the records by name (tests and tools read them — and substitute one, e.g. a misaligned coordinate table; the
    # render paths take all three from one _receiver() call)
    _plane = property(lambda self: self._receiver()[0], lambda self, v: self._override_record(0, v))
    _xs = property(lambda self: self._receiver()[1], lambda self, v: self._override_record(1, v))
    _ys = property(lambda self: self._receiver()[2], lambda self, v: self._override_record(2, v))

    @property
    def device_trig(self) -> bool:
        return self._device_trig

    @device_trig.setter
    def device_trig(self, value: bool) -> None:
        self._device_trig = bool(value)
        self._trig_cache.clear()            # tables of the other kind are not reused

    # ------------------------------------------------------------------ errors
    def reset_errors(self) -> None:
        """Re-draw both error tensors (:220-239); same RNG call order as the reference."""
        self.error_angles_mrad = (
            torch.randn(self.num_heliostats, 2, device=self.device) * self.error_scale_mrad)
        if self.max_batch_size >= 1:
            self.batch_error_angles_mrad = self._sample_error_angles(self.max_batch_size)
        else:
            self.batch_error_angles_mrad = None
        # the trig tables of the new errors, now: the host round trip of device-sampled errors belongs to
        # reset_errors(), not to the first render after it (which may be under graph capture) — and it is ONE
        # round trip for both tensors (HelioEnv.reset() runs this every few steps: test_environment.py:386)
        self._prime_trig_tables()

    def _prime_trig_tables(self) -> None:
        single, batch = self.error_angles_mrad, self.batch_error_angles_mrad
        if batch is None or not single.is_cuda or self.device_trig:
            self._cached_trig("single", single)
            if batch is not None:
                self._cached_trig("batch", batch)
            return
        N = self.num_heliostats
        both = self._trig_of(torch.cat([single.reshape(1, N, 2), batch], dim=0))      # [1 + max_batch, N, 4]
        for slot, errs, table in (("single", single, both[0]), ("batch", batch, both[1:])):
            key = (errs.data_ptr(), errs._version, tuple(errs.shape), errs.device)
            self._trig_cache[slot] = (key, table, errs)

    def _sample_error_angles(self, batch_size: int) -> torch.Tensor:
        """[batch_size, N, 2] fresh error angles in mrad (:243-252)."""
        return torch.randn(batch_size, self.num_heliostats, 2, device=self.device) * self.error_scale_mrad

    def _trig_of(self, errs: torch.Tensor, on_device: bool = False) -> torch.Tensor:
        """(cos_e, sin_e, cos_u, sin_u) of errs·1e-3 (:87-91) WITH TORCH'S CPU BITS, wherever ``errs``
        lives: the 1e-5 image tolerance at sigma_scale = 0.01 leaves no room for a 1-ulp-different
        trig table, so device-sampled errors make one round trip per
        ``reset_errors()`` — download, torch's CPU ``cos``/``sin`` (the kernels the reference's CPU
        path runs), upload — and the table is cached until the tensor changes.  ``device_trig =
        True`` (or HELIO_DEVICE_TRIG=1) keeps everything on the device instead
        (``helio_error_trig``: ≤ 1 ulp from these values, no host synchronisation)."""
        if errs.is_cuda and (self.device_trig or on_device):
            return _get_ops().error_trig(errs).to(self.device)
        if errs.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError(
                "HelioField: the cos/sin table of device-resident error angles is made with torch's CPU kernels (the "
                "reference's bits) — a device→host→device round trip that cannot be captured in a HIP graph.  Call "
                "reset_errors() / assign the error tensors BEFORE the capture (the table is cached until they change), "
                "or set field.device_trig = True (all-device table, ≤ 1 ulp from the CPU one)")
        ang = errs.detach().to(device="cpu", dtype=torch.float32) * 1e-3
        e, u = ang[..., 0], ang[..., 1]
        t = torch.stack([e.cos(), e.sin(), u.cos(), u.sin()], dim=-1)
        return t.to(self.device).contiguous()

    def _cached_trig(self, slot: str, errs: torch.Tensor) -> torch.Tensor:
        hit = self._trig_cache.get(slot)
        # same tensor object, not written to since: the cheap test first (this runs on every render)
        if hit is not None and hit[2] is errs and hit[0][1] == errs._version:
            return hit[1]
        key = (errs.data_ptr(), errs._version, tuple(errs.shape), errs.device)
        if hit is None or hit[0] != key:
            hit = (key, self._trig_of(errs), errs)     # keep errs alive: its data_ptr is the key
            self._trig_cache[slot] = hit
        return hit[1]

    def _select_trig(self, B: int, row0: int = 0, rows: int | None = None):
        """Error-selection rule of render() (:340-353) → (trig table, batch stride).

        ``B`` is the size of the (global) batch the rule is applied to; a sharded render
        passes the rows ``row0 : row0+rows`` it owns and gets exactly those rows of the
        table the unsharded call would use."""
        N = self.num_heliostats
        rows = B if rows is None else rows
        if B == 1:
            return self._cached_trig("single", self.error_angles_mrad), 0
        batch = self.batch_error_angles_mrad
        if batch is not None and B <= batch.shape[0]:
            table = self._cached_trig("batch", batch)           # rows [:B] are a prefix
            return (table if row0 == 0 else table[row0:row0 + rows]), 4 * N
        # more suns than pre-sampled errors: the reference draws fresh errors on EVERY call (:349-353).  Their
        # cos/sin carry torch's CPU bits like every other table (a 1-ulp-different table is 9e-6…1.7e-5 of the
        # peak at sigma_scale = 0.01, SURVEY App. B — the whole 1e-5 budget): one host round trip per render
        # on this branch, which HelioEnv never takes (max_batch_size = batch_size); field.device_trig = True
        # keeps it on the device.  Pinned by test_fresh_error_branch_holds_the_1e5_bar (seeded device draw).
        return self._trig_of(self._sample_error_angles(rows)), 4 * N

    # ------------------------------------------------------------------ optics
    def calculate_ideal_normals(self, sun_position) -> torch.Tensor:
        """Normals that send every heliostat's reflection to the target centre (:256-278)."""
        sun = torch.as_tensor(sun_position, dtype=torch.float32, device=self.device)
        self._receiver()                                  # (target_position as it is NOW, :263 / :274)
        if sun.dim() == 1:
            return _get_ops().ideal_normals(self.heliostat_positions, sun.view(1, 3).contiguous(),
                                            self._target_xyz)[0]
        return _get_ops().ideal_normals(self.heliostat_positions, sun.contiguous(), self._target_xyz)

    def init_actions(self, sun_position) -> None:
        """Noisy initial mirror orientations (:291-304); always consumes one randn_like."""
        self._init_actions_from(self.calculate_ideal_normals(sun_position))

    def _init_actions_from(self, ideal: torch.Tensor) -> None:
        """``init_actions`` given the ideal normals of the sun position(s) (HelioEnv.reset has them
        cached: they depend only on geometry the two fields share)."""
        noisy = _get_ops().init_actions(ideal, torch.randn_like(ideal), self.initial_action_noise)
        self.initial_action = noisy.flatten() if ideal.dim() == 2 else noisy.view(ideal.shape[0], -1)

    def render(self, sun_position, action, ideal_normals=None, show_spillage: bool = False,
               monitor: bool = False):
        """Flux image(s) on the receiver (:308-415).

        Returns ``(image, actual)`` or, with ``monitor``, ``(image, actual, refl)``;
        ``image`` is ``[R,R]`` for a 1-D sun and ``[B,R,R]`` otherwise, ``actual`` is
        ``[B,N,3]`` (``[1,N,3]`` for a 1-D sun), ``refl`` is ``[B·N,3]``.
        ``ideal_normals`` and ``show_spillage`` are accepted and unused, as in the reference.
        """
        memo = self._fast
        if memo is not None:
            # the context of the previous call, still valid (every assignment to an attribute of this field
            # clears the memo; in-place writes to the error tensor or to one of the receiver's tensors, a forced
            # variant and gradient recording are checked inside): argument checks, allocation, launch and the
            # reference's return shapes in one compiled call — at config 2 the kernel needs 3.7 µs, so every host
            # microsecond shows
            out = memo.render_checked(sun_position, action, monitor)
            if out is not None:
                return out
        picked = None        # the error table once selected for this call: the fresh-error branch draws from the RNG
        if (type(sun_position) is torch.Tensor and type(action) is torch.Tensor
                and not (action.requires_grad and torch.is_grad_enabled())):
            # launch-bound path: dtype / device / shape fix-ups, allocation and the launch happen inside the compiled
            # binding's render context (None without it)
            batched = sun_position.dim() > 1
            B = sun_position.shape[0] if batched else 1
            ctx, trig, stride, cached = self._render_context(B)
            picked = (trig, stride)
            out = ctx.render(sun_position if batched else sun_position.unsqueeze(0), action, monitor) if ctx is not None else None
            if out is not None:
                if cached:                               # the next call may skip all of the above
                    ctx.bind_errors(self.error_angles_mrad if stride == 0 else self.batch_error_angles_mrad)
                    d = self.__dict__
                    ctx.bind_receiver([d["_target_position"], d["_target_normal"], d["_plane_u"], d["_plane_v"]])
                    self._fast = ctx
                if not monitor:
                    return out if batched else (out[0][0], out[1])
                img = out[0] if batched else out[0][0]
                return img, out[1], out[2].view(-1, 3)
        sun = torch.as_tensor(sun_position, dtype=torch.float32, device=self.device)
        batched = sun.dim() > 1
        if not batched:
            sun = sun.unsqueeze(0)
        # (a call that got as far as selecting its errors above and then needs the general path — ctypes binding,
        # an action that needs a dtype fix-up — keeps that selection: ONE draw per render, as in the reference :349-353)
        img, actual, refl = self.render_rows(sun, action, 0, sun.shape[0], monitor, _picked=picked)
        if not batched:
            img = img[0]
        return (img, actual, refl) if monitor else (img, actual)

    def _render_context(self, B: int):
        """→ (compiled render context or None, trig table, batch stride, cached) for a batch of ``B`` suns — the ONE
        place that decides whether the context of an earlier call still serves: it is rebuilt when the errors, the
        forced variant, the binding, the receiver (a new plane record / pixel grid: any of its attributes assigned or
        written in place) or the heliostat tensor change.  ``cached``: the trig table belongs to stored errors (False:
        drawn for this call, :349-353 — nothing to memoise)."""
        fast = self._fast_render
        if fast is None:
            self._ops = _get_ops()
            fast = self._fast_render = getattr(self._ops, "render_context", False)
        trig, stride = self._select_trig(B)
        batch = self.batch_error_angles_mrad
        cached = B == 1 or (batch is not None and B <= batch.shape[0])
        plane, xs, _ = self._receiver()
        if not fast:
            return None, trig, stride, cached
        ops, key = self._ops, self._ctx_key
        if (key is None or key[0] is not trig or key[1] != stride or key[2] != ops.splat_variant
                or key[3] is not ops.hb or key[4] is not plane or key[5] is not self.heliostat_positions or key[6] is not xs):
            self._render_ctx = fast(self, trig, stride)
            self._ctx_key = (trig, stride, ops.splat_variant, ops.hb, plane, self.heliostat_positions, xs)
        return self._render_ctx, trig, stride, cached

    def render_value_and_grad(self, sun_position, action, grad_image=None, grad_actual=None, grad_refl=None):
        """``render`` and the gradient of a scalar loss w.r.t. ``action`` in one call, for GIVEN cotangents
        ``dL/dimage [B,R,R]``, ``dL/dactual [B,N,3]``, ``dL/drefl [B·N,3]`` (any may be None): the forward
        kernels and the backward kernels are enqueued back to back, with no autograd graph in between —
        the same numbers as ``render`` + ``torch.autograd.grad`` (GPU test), at a fraction of the host
        time (BASELINE config 3).  → ``(image, actual, grad_action [B, 3N])`` with the shapes of
        :meth:`render`; the inputs are treated as constants (no graph is recorded)."""
        sun = torch.as_tensor(sun_position, dtype=torch.float32, device=self.device)
        batched = sun.dim() > 1
        if not batched:
            sun = sun.unsqueeze(0)
        sun = sun.contiguous()
        B, N = sun.shape[0], self.num_heliostats
        act = torch.as_tensor(action, dtype=torch.float32, device=self.device).detach().reshape(B, N, 3).contiguous()
        fix = lambda g, shape: None if g is None else torch.as_tensor(  # noqa: E731
            g, dtype=torch.float32, device=self.device).detach().reshape(shape).contiguous()
        g_img = fix(grad_image, (B, self.resolution, self.resolution))
        g_act, g_refl = fix(grad_actual, (B, N, 3)), fix(grad_refl, (B, N, 3))
        ctx, trig, stride, _ = self._render_context(B)
        out = ctx.render_and_grad(sun, act, g_img, g_act, g_refl, self._ops.bwd_variant) if ctx is not None else None
        if out is None:
            ops = _get_ops()
            plane, xs, ys = self._receiver()
            ws = torch.empty((B, N, native.RAY_STRIDE), dtype=torch.float32, device=act.device)
            image, actual, _, rays = ops.render_fwd(self.heliostat_positions, sun, act, trig, stride, plane,
                                                    xs, ys, want_refl=False, rays=ws)
            grad = ops.render_bwd(self.heliostat_positions, sun, act, trig, stride, plane, rays, xs,
                                  ys, g_img, g_act, g_refl)
            out = (image, actual, grad)
        image, actual, grad = out
        return (image if batched else image[0]), actual, grad.view(B, -1)

    def render_rows(self, sun_rows, action_rows, row_offset: int, global_batch: int, monitor: bool = False, _picked=None):
        """Render rows ``row_offset : row_offset+len(sun_rows)`` of a batch of
        ``global_batch`` suns (the whole batch when called by :meth:`render`; one shard of
        it when called by :class:`doodle_amd.sharded.ShardedRenderer`).  The error rows are
        those the unsharded render would use for the same suns.  Always returns
        ``(images [b,R,R], actual [b,N,3], refl [b·N,3] or None)``."""
        sun = torch.as_tensor(sun_rows, dtype=torch.float32, device=self.device).contiguous()
        B, N = sun.shape[0], self.num_heliostats

        act = torch.as_tensor(action_rows, dtype=torch.float32, device=self.device)
        normals = act.reshape(B, N, 3).contiguous()
        trig, stride = _picked if _picked is not None else self._select_trig(global_batch, row_offset, B)

        # a piece of a larger batch is rendered by the kernels the WHOLE batch would get (the size rules look at
        # B; every kernel's summation order depends on N and R only): the rows of the unsharded render AND of its
        # gradient, bit for bit, whatever the shard size
        # — passed DOWN with this call (arguments of helio_render_fwd / helio_render_bwd), never written to the
        # process-wide ops object: another thread rendering meanwhile keeps its own kernel choice.
        ops = _get_ops()
        plane, xs, ys = self._receiver()
        forced = forced_bwd = None
        if global_batch != B:
            if getattr(ops, "splat_variant", 0) == 0:
                choose = getattr(ops, "render_choice", None)
                forced = (choose(global_batch, N, self.resolution) or None) if choose is not None else None
            if getattr(ops, "bwd_variant", 0) == 0:
                choose = getattr(ops, "render_bwd_choice", None)
                forced_bwd = (choose(global_batch, N, self.resolution) or None) if choose is not None else None
        if torch.is_grad_enabled() and normals.requires_grad:
            node = getattr(ops, "render_node", None)
            out = node(self, sun, normals, trig, stride, variant=forced, bwd_variant=forced_bwd) if node is not None else None
            # (the same node as a C++ autograd Function when the compiled binding is built)
            images, actual, refl = out if out is not None else _Render.apply(normals, self, sun, trig, stride, forced, forced_bwd)
        else:
            # no autograd: the ray work buffer is scratch, reuse it between calls
            ws = self._ray_ws
            if ws is None or ws.shape[0] != B or ws.device != normals.device:
                ws = self._ray_ws = torch.empty((B, N, native.RAY_STRIDE), dtype=torch.float32, device=normals.device)
            images, actual, refl, _ = ops.render_fwd(
                self.heliostat_positions, sun, normals, trig, stride, plane, xs, ys,
                want_refl=monitor, rays=ws, variant=forced)
        return images, actual, (refl.view(-1, 3) if refl is not None else None)
