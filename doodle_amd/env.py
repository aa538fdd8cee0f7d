"""HelioEnv — the reference's Gym-style environment on top of the HIP render path.

Mirrors ``HelioEnv`` of DOODLE's ``test_environment.py`` (:175-525): constructor
signature, ``reset()`` / ``step()`` / ``set_sun_pos()`` / ``seed()``, the observation and
metric dictionaries, and the attributes the training scripts read
(``train_with_env.py:171-216``).  The two ``HelioField`` renders and the ideal-normal
computation per step run in libhelio.so, and so do the loss block (:427-457), the boundary
loss (:101-130), the alignment angle (:132-155) and the distance maps (:92-97) — SURVEY.md §8(f).

``gymnasium`` is optional: when it is absent the spaces are small records with the
same fields (the reference only stores them, ``test_environment.py:241-252``).
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import field as _field
from .field import HelioField
from .losses import StepConstants, env_step_fused

try:  # pragma: no cover - depends on the image
    import gymnasium as _gym
    from gymnasium import spaces as _spaces
    _EnvBase = _gym.Env
    _Box, _DictSpace = _spaces.Box, _spaces.Dict
except Exception:  # gymnasium is not installed in this image
    class _EnvBase:  # noqa: D401
        """Stand-in for gymnasium.Env (no behaviour is inherited by the reference env)."""

    class _Box:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class _DictSpace(dict):
        @property
        def spaces(self):
            return self


SUN_RANGE = math.hypot(10000, 10000)   # :324


def direction_from_azimuth_elevation(azimuth_deg: float, elevation_deg: float, device=None) -> torch.Tensor:
    """Unit vector for an azimuth (0° = +X, CCW towards +Y) and elevation (:18-40)."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    d = torch.tensor([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)],
                     dtype=torch.float32, device=device)
    return d / torch.norm(d)


azimuth_elevation_to_primary_direction = direction_from_azimuth_elevation   # the reference's name


def sample_cone_directions(n: int, axis: torch.Tensor, half_angle_deg: float, device=None,
                           force_upper_hemisphere: bool = False) -> torch.Tensor:
    """``n`` unit vectors uniform on the spherical cap around ``axis`` (:42-88).
    Draws ``rand(n)`` twice, in the reference's order (cos θ first, then φ)."""
    device = device or axis.device
    a = F.normalize(axis.to(device), dim=0)
    helper = torch.tensor([0.0, 0.0, 1.0], device=device)
    if torch.abs(a[2]) > 0.999:
        helper = torch.tensor([0.0, 1.0, 0.0], device=device)
    e1 = F.normalize(torch.linalg.cross(helper, a), dim=0)
    e2 = torch.linalg.cross(a, e1)
    cos_t = 1.0 - torch.rand(n, device=device) * (1.0 - math.cos(math.radians(half_angle_deg)))
    sin_t = torch.sqrt(torch.clamp(1.0 - cos_t ** 2, min=0.0))
    phi = 2.0 * math.pi * torch.rand(n, device=device)
    dirs = (e1[None, :] * (sin_t * torch.cos(phi))[:, None]
            + e2[None, :] * (sin_t * torch.sin(phi))[:, None]
            + a[None, :] * cos_t[:, None])
    dirs = F.normalize(dirs, dim=1)
    if force_upper_hemisphere:
        dirs[:, 2] = torch.abs(dirs[:, 2])
    return dirs


def make_distance_maps(imgs: torch.Tensor, thr: float = 0.5) -> torch.Tensor:
    """Euclidean distance to the region above ``thr``·max of each image (:92-97): the exact
    device EDT of csrc/edt.hip (equal to scipy's ``distance_transform_edt`` bit for bit) —
    no host round trip, and no CPU implementation in the product."""
    from . import field as _field
    return _field._get_ops().distance_maps(imgs.detach().to(torch.float32), float(thr))


class HelioEnv(_EnvBase):
    """Batched heliostat-aiming environment (reference: test_environment.py:175)."""

    def __init__(self, heliostat_pos, targ_pos, targ_area, targ_norm, sigma_scale=0.1,
                 error_scale_mrad=180.0, initial_action_noise=0.0, resolution=128, batch_size=25,
                 device="cuda", new_sun_pos_every_reset=False, new_errors_every_reset=True,
                 use_error_mask=False, error_mask_ratio=0.2, exponential_risk=False,
                 single_sun=False, azimuth=45.0, elevation=45.0):
        super().__init__()
        as_t = lambda x: x if isinstance(x, torch.Tensor) else torch.tensor(x, dtype=torch.float32, device=device)  # noqa: E731
        heliostat_pos, targ_pos, targ_norm = as_t(heliostat_pos), as_t(targ_pos), as_t(targ_norm)

        self.resolution, self.batch_size, self.device = resolution, batch_size, device
        self.heliostat_pos = heliostat_pos
        self.num_heliostats = heliostat_pos.shape[0]
        self.targ_pos, self.targ_area, self.targ_norm = targ_pos, targ_area, targ_norm
        self.azimuth, self.elevation = azimuth, elevation
        self.sigma_scale, self.error_scale_mrad = sigma_scale, error_scale_mrad
        self.initial_action_noise = initial_action_noise
        self.sun_pos = None
        self.sun_errors = None
        self.new_sun_pos_every_reset = new_sun_pos_every_reset
        self.new_errors_every_reset = new_errors_every_reset
        self.single_sun = single_sun
        self.use_error_mask, self.error_mask_ratio = use_error_mask, error_mask_ratio
        self.exponential_risk = exponential_risk
        # the NaN/Inf asserts of :495-501, one host sync per step (False skips it; measured: the step
        # is then bound by its three-kernel GPU chain, 34 µs instead of 48 µs at config 2)
        # True: the reference's NaN/Inf asserts after every step (:495-501); False: none; "deferred": the
        # same asserts one step late (finish_checks() for the last one) — the host never waits for the GPU
        self.check_finite = True
        self._pending_check = None
        self._consts_cache = None
        self._step_ctx, self._step_ctx_key = None, None
        self._ref_cache = None
        f3 = ctypes.c_float * 3
        self._tp3 = f3(*[float(x) for x in targ_pos.detach().cpu().tolist()])
        self._tn3 = f3(*[float(x) for x in targ_norm.detach().cpu().tolist()])

        n_act = self.num_heliostats * 3
        self.action_space = _Box(low=-1.0, high=1.0, shape=(n_act,), dtype=np.float32)
        self.observation_space = _DictSpace({
            "img": _Box(low=0.0, high=np.inf, shape=(batch_size, resolution, resolution), dtype=np.float32),
            "aux": _Box(low=-np.inf, high=np.inf, shape=(batch_size, 3 + n_act), dtype=np.float32),
        })

        # error-free reference field first, then the noisy one (RNG order of :255-277)
        common = dict(heliostat_positions=heliostat_pos, target_position=targ_pos, target_area=targ_area,
                      target_normal=targ_norm, sigma_scale=sigma_scale, resolution=resolution,
                      max_batch_size=batch_size, device=device)
        self.ref_field = HelioField(error_scale_mrad=0.0, **common)
        self.noisy_field = HelioField(error_scale_mrad=error_scale_mrad, **common)

        self.set_sun_pos(self._draw_sun_directions() * SUN_RANGE)

    # ------------------------------------------------------------------ suns
    def _draw_sun_directions(self) -> torch.Tensor:
        """:286-321 — a 2° cone around (azimuth, elevation), or the whole upper hemisphere."""
        n = 1 if self.single_sun else self.batch_size
        if self.azimuth is not None and self.elevation is not None:
            axis = direction_from_azimuth_elevation(self.azimuth, self.elevation, device=self.device)
            dirs = sample_cone_directions(n, axis, 2.0, device=self.device, force_upper_hemisphere=True)
            return dirs.repeat(self.batch_size, 1) if self.single_sun else dirs
        dirs = F.normalize(torch.randn(n, 3, device=self.device), dim=1)
        if self.single_sun:
            dirs = dirs.repeat(self.batch_size, 1)
        dirs[:, 2] = torch.abs(dirs[:, 2])
        return dirs

    def _reference(self):
        """Ideal normals, the error-free field's image of them and its per-image peak
        (:414, :429-436).  The reference recomputes all three on every step although they
        depend only on the sun positions and the (zero) reference errors; here they are
        cached until either changes."""
        errs = self.ref_field.batch_error_angles_mrad
        single = self.ref_field.error_angles_mrad
        plane, xs, _ = self.ref_field._receiver()      # new records whenever a receiver attribute of the field changed
        hit = self._ref_cache
        if hit is not None:
            # the same tensor objects, not written to since (this runs on every step): cheap test first
            sun0, sv, e0, ev, s0, ssv, p0, x0, h0 = hit[6]
            if (sun0 is self.sun_pos and sv == sun0._version and e0 is errs and (errs is None or ev == errs._version)
                    and s0 is single and ssv == single._version and p0 is plane and x0 is xs
                    and h0 is self.ref_field.heliostat_positions):
                return hit[1:5]
        key = (self.sun_pos.data_ptr(), self.sun_pos._version,
               None if errs is None else (errs.data_ptr(), errs._version), single.data_ptr(), single._version,
               id(plane), id(xs), self.ref_field.heliostat_positions.data_ptr())
        if self._ref_cache is None or self._ref_cache[0] != key:
            with torch.no_grad():
                ideal = self.ref_field.calculate_ideal_normals(self.sun_pos)
                target, _ = self.ref_field.render(self.sun_pos, ideal.flatten(1), ideal)
                tx = target.amax((1, 2)).clamp_min(1e-6)
            self._ref_cache = (key, ideal, target, tx, ideal.view([-1, 3]), (errs, single, plane, xs), None)
        self._ref_cache = self._ref_cache[:6] + ((self.sun_pos, self.sun_pos._version, errs,
                                                  None if errs is None else errs._version, single, single._version,
                                                  plane, xs, self.ref_field.heliostat_positions),)
        return self._ref_cache[1:5]

    def set_sun_pos(self, sun_positions: torch.Tensor):
        """Fix the sun positions and precompute the reference image statistics (:359-370)."""
        self.sun_pos = sun_positions.clone().detach()
        self._ref_cache = None
        self.ref_field.init_actions(self.sun_pos)
        with torch.no_grad():
            ideal = self.ref_field.calculate_ideal_normals(self.sun_pos)
            timg, _ = self.ref_field.render(self.sun_pos, self.ref_field.initial_action, ideal)
        self.distance_maps = make_distance_maps(timg)
        self.ref_min = torch.min(timg)
        self.ref_max = torch.max(timg)

    # ------------------------------------------------------------------ gym API
    def reset(self):
        """→ ``{'img': [B,R,R], 'aux': [B, 3+3N]}`` (:372-400)."""
        if self.new_sun_pos_every_reset:
            # the reference branch (:378-385) calls an undefined method and cannot run
            raise NotImplementedError("new_sun_pos_every_reset=True is broken in the reference "
                                      "(test_environment.py:378-385); call set_sun_pos() instead")
        if self.new_errors_every_reset:
            self.noisy_field.reset_errors()
        with torch.no_grad():
            # the reference recomputes the ideal normals twice here (:386-392); they are the cached ones
            ideal = self._reference()[0]
            self.noisy_field._init_actions_from(ideal)
            img, _ = self.noisy_field.render(self.sun_pos, self.noisy_field.initial_action, ideal)
        self.ideal_normals = ideal
        return {"img": img, "aux": torch.cat([self.sun_pos, ideal.flatten(1)], dim=1)}

    def __getstate__(self):
        # compiled step contexts and caches keyed on object identity are rebuilt on demand
        state = dict(self.__dict__)
        for name in ("_step_ctx", "_step_ctx_key", "_consts_cache", "_pending_check", "_ref_cache"):
            if name in state:
                state[name] = None
        return state

    def _run_step(self, action, consts):
        """The step's render + loss block: → ((img, actual, refl [B·N,3], mse, dist, bound, alignment_loss, flag,
        mae [B,1], angles [B·N], all_bounds [B,N], aux, normals [B,N,3]), ticket).

        The ONE place that decides how ``helio_env_step_fwd`` is reached — every route is the same C call with the same
        arguments and gives the same bits (tests/test_gpu_more.py::test_env_step_routes_agree):
          1. the compiled step context (everything constant between steps bound once; per step two tensors and a
             ticket), as one autograd node when the action records a gradient;
          2. the compiled binding's unbound entry, for an action that needs a dtype / device / layout fix-up;
          3. without the compiled binding: ``losses.env_step_fused`` (the same node as a Python autograd Function) or
             the ctypes call.
        A new invalidation rule goes into the context key below and nowhere else."""
        nf, ops = self.noisy_field, _field._get_ops()
        differentiate = torch.is_grad_enabled() and action.requires_grad
        notify = bool(self.check_finite)
        trig, stride = nf._select_trig(self.batch_size)
        plane, xs, ys = nf._receiver()
        make_ctx = getattr(ops, "env_step_context", None)
        if make_ctx is not None and type(action) is torch.Tensor:
            key = self._step_ctx_key
            if (key is None or key[0] is not consts or key[1] is not trig or key[2] != ops.splat_variant
                    or key[3] is not ops.hb or key[4] is not plane or key[5] is not nf.heliostat_positions
                    or key[6] is not xs):
                # constants, errors, forced kernel variant, binding, the receiver (a new plane record / pixel
                # grid) or the heliostat tensor changed: rebind the step context
                self._step_ctx = make_ctx(nf, trig, stride, consts)
                self._step_ctx_key = (consts, trig, ops.splat_variant, ops.hb, plane, nf.heliostat_positions, xs)
            ctx = self._step_ctx
            if ctx is not None:                                          # route 1
                ticket = ops.next_ticket() if notify else 0
                out = (ctx.step_grad(self.sun_pos, action, ops.bwd_variant, ticket) if differentiate
                       else ctx.step(self.sun_pos, action, ticket))
                if out is not None:
                    return out, ticket
            if not differentiate:                                        # route 2
                out = ops.env_step_nograd(nf, self.sun_pos, action, trig, stride, consts, notify=notify)
                if out is not None:
                    return out[:-1], out[-1]
        normals = action.view(self.batch_size, -1, 3)                    # :460          (route 3)
        if differentiate:
            (img, actual, reflected, mse, dist_l, bound, alignment_loss, mae, angles, all_bounds,
             flag, ticket) = env_step_fused(nf, self.sun_pos, normals.contiguous(), consts, notify=notify)
        else:
            n3 = torch.as_tensor(action, dtype=torch.float32, device=self.device).detach().reshape(
                self.batch_size, -1, 3).contiguous()
            (img, actual, reflected, _rays, out5, mae, angles, all_bounds, _keep, _aux, ticket) = ops.env_step_fwd(
                nf.heliostat_positions, self.sun_pos, n3, trig, stride, plane, xs, ys, consts, notify=notify)
            mse, dist_l, bound, alignment_loss, flag = out5[0], out5[1], out5[2], out5[3], out5[4]
        aux = torch.cat([self.sun_pos.detach(), action.flatten(1)], dim=1)
        return (img, actual, reflected.view([-1, 3]), mse, dist_l, bound, alignment_loss, flag, mae.view([-1, 1]),
                angles.view([-1]), all_bounds, aux, normals), ticket

    def step(self, action):
        """Render ``action`` on the noisy field and score it (:402-516).

        Returns ``(obs, metrics, monitor)`` with the reference's keys; gradients flow to
        ``action`` through ``img`` (mse, dist) and ``actual`` (alignment_loss).
        """
        if isinstance(action, np.ndarray):
            action = torch.tensor(action, dtype=torch.float32, device=self.device)
        if self.use_error_mask and self.batch_size > 4096:
            raise NotImplementedError("use_error_mask: the fused quantile covers batch_size <= 4096")
        ideal, target, tx, ideal_flat = self._reference()
        mask_ratio = float(self.error_mask_ratio) if self.use_error_mask else -1.0
        cc = self._consts_cache
        if (cc is not None and cc[0] is target and cc[1] is self.distance_maps and cc[2] == self.exponential_risk
                and cc[3] == mask_ratio and cc[4].helios is self.noisy_field.heliostat_positions):
            consts = cc[4]                                               # nothing the loss block reads has changed
        else:
            consts = StepConstants(target, tx, self.distance_maps, ideal, self.noisy_field.heliostat_positions,
                                   self._tp3, self._tn3, float(self.targ_area[0]), float(self.targ_area[1]),
                                   bool(self.exponential_risk), mask_ratio)
            self._consts_cache = (target, self.distance_maps, self.exponential_risk, mask_ratio, consts)
        (img, actual, reflected, mse, dist_l, bound, alignment_loss, flag, mae, angles, all_bounds, aux,
         normals), ticket = self._run_step(action, consts)
        metrics = {"mse": mse, "dist": dist_l, "bound": bound, "alignment_loss": alignment_loss}
        obs = {"img": img, "aux": aux}
        monitor = {
            "normals": normals,
            "reflected_rays": reflected,
            "ideal_normals": ideal_flat,
            "all_bounds": all_bounds,
            "mae_image": mae,
            "alignment_errors": angles,
        }
        if self.check_finite:                                          # :495-501, one wait instead of six syncs;
            # last, so that the dictionaries above are built while the GPU finishes the step.  The
            # finishing workgroup publishes the flag to pinned host memory (helio_notify_*)
            if self.check_finite == "deferred":
                # rollouts: this step's flag is looked at when the NEXT step (or finish_checks()) comes,
                # by which time it has long been published — the host never waits for the GPU, and a
                # NaN/Inf is reported one step late
                pending, self._pending_check = self._pending_check, (ticket, flag)
                if pending is not None:
                    self._raise_if_nonfinite(*pending)
            else:
                self._raise_if_nonfinite(ticket, flag)
        return obs, metrics, monitor

    def _raise_if_nonfinite(self, ticket, flag):
        bad = _field._get_ops().notify_wait(ticket) if ticket else None
        if bool(flag) if bad is None else bad:
            raise AssertionError("MSE, distance loss or boundary loss is NaN or Inf")

    def finish_checks(self):
        """With ``check_finite = "deferred"``: look at the flag of the last step now."""
        pending, self._pending_check = self._pending_check, None
        if pending is not None:
            self._raise_if_nonfinite(*pending)

    def seed(self, seed=None):
        """Seed torch and numpy (:518-525)."""
        if seed is not None:
            torch.manual_seed(seed)
            np.random.seed(seed)
