"""Fused loss block of HelioEnv.step on the HIP kernels of csrc/step_losses.hip.

Replaces, for the ``use_error_mask=False`` branch, the ≈60 small torch launches of the
reference's ``step`` (test_environment.py:436-488) by two launches forward and one backward
(C ABI: ``helio_step_losses_fwd`` / ``helio_step_losses_bwd``).  Gradients flow to ``img``
(mse, dist), ``actual`` (alignment_loss) and ``action`` (bound), exactly the edges autograd
builds in the reference.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from . import field as _field


@dataclass
class StepConstants:
    """Everything of the loss block that does not depend on the action."""
    target: torch.Tensor      # [B,R,R]  reference-field image of the ideal normals (detached)
    tx: torch.Tensor          # [B]      clamp_min(amax(target[b]), 1e-6)
    dmaps: torch.Tensor       # [B,R,R]  distance maps
    ideal: torch.Tensor       # [B,N,3]
    helios: torch.Tensor      # [N,3]
    tp: ctypes.Array          # target position, float[3]
    tn: ctypes.Array          # target normal as the env holds it, float[3]
    W: float
    H: float
    exp_risk: bool
    mask_ratio: float = -1.0  # use_error_mask: fraction of worst images kept (< 0: no mask)

    def __post_init__(self):
        # what the compiled binding takes, converted once (the constants object is reused across steps)
        self.tp_l, self.tn_l = list(self.tp), list(self.tn)
        self.exp_risk, self.mask_ratio = bool(self.exp_risk), float(self.mask_ratio)


class _StepLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, actual, action, consts):
        out, mae, align, allb, keep = _field._get_ops().step_losses_fwd(img, actual, action, consts)
        ctx.consts = consts
        ctx.save_for_backward(img, actual, action, keep)
        flag = out[4]
        ctx.mark_non_differentiable(mae, align, allb, flag)
        return out[0], out[1], out[2], out[3], mae, align, allb, flag

    @staticmethod
    def backward(ctx, g_mse, g_dist, g_bound, g_align, *_unused):
        img, actual, action, keep = ctx.saved_tensors
        want = ctx.needs_input_grad
        c = lambda g: g.contiguous() if g is not None else None  # noqa: E731
        gi, ga, gn = _field._get_ops().step_losses_bwd(
            img, actual, action, ctx.consts, c(g_mse), c(g_dist), c(g_bound), c(g_align), keep,
            want[0] and (g_mse is not None or g_dist is not None), want[1] and g_align is not None,
            want[2] and g_bound is not None)
        return gi, ga, gn, None


class _EnvStep(torch.autograd.Function):
    """render + loss block as ONE autograd node (what HelioEnv.step differentiates): forward =
    helio_env_step_fwd (render + loss block; 2 launches for small problems), backward = helio_env_step_bwd (or, with an external image cotangent, helio_step_losses_bwd + helio_render_bwd).
    Same kernels as ``_Render`` followed by ``_StepLosses``; it only spares the autograd engine
    five graph nodes per step (the TTT inner loop of the reference calls step+backward 800 times
    per optimiser step)."""

    @staticmethod
    def forward(ctx, normals, field, sun, trig, trig_stride, consts, notify):
        ops = _field._get_ops()
        plane, xs, ys = field._receiver()
        image, actual, refl, rays, out, mae, align, allb, keep, _, ticket = ops.env_step_fwd(
            field.heliostat_positions, sun, normals, trig, trig_stride, plane, xs, ys, consts,
            notify=notify)
        ctx.trig_stride, ctx.consts = trig_stride, consts
        ctx.scene = (field.heliostat_positions, plane, xs, ys)      # as stepped: later assignments to the field do not reach this node
        ctx.save_for_backward(normals, sun, trig, rays, image, actual, keep)
        ctx.set_materialize_grads(False)
        flag = out[4]
        ctx.mark_non_differentiable(mae, align, allb, flag)
        return image, actual, refl, out[0], out[1], out[2], out[3], mae, align, allb, flag, ticket

    @staticmethod
    def backward(ctx, g_image, g_actual, g_refl, g_mse, g_dist, g_bound, g_align, *_unused):
        normals, sun, trig, rays, image, actual, keep = ctx.saved_tensors
        ops = _field._get_ops()
        helios, plane, xs, ys = ctx.scene
        c = lambda g: g.contiguous() if g is not None else None  # noqa: E731
        step_bwd = getattr(ops, "env_step_bwd", None)
        if g_image is None and step_bwd is not None:
            # the whole backward in one C call (helio_env_step_bwd): 1-3 launches, no temporaries
            # for the ray-loss adjoints, no [B,R,R] image cotangent when there are few rays
            g = step_bwd(helios, sun, normals, trig, ctx.trig_stride, plane, rays,
                         xs, ys, image, ctx.consts, c(g_mse), c(g_dist), c(g_bound), c(g_align), keep,
                         c(g_actual), c(g_refl))
            return g, None, None, None, None, None, None
        need_img = g_mse is not None or g_dist is not None
        gi = ga = gn = None
        if need_img or g_align is not None or g_bound is not None:
            gi, ga, gn = ops.step_losses_bwd(image, actual, normals, ctx.consts, c(g_mse), c(g_dist), c(g_bound),
                                             c(g_align), keep, need_img, g_align is not None, g_bound is not None)
        g_image = gi if g_image is None else (g_image if gi is None else g_image + gi)
        g_actual = ga if g_actual is None else (g_actual if ga is None else g_actual + ga)
        if g_image is None and g_actual is None and g_refl is None:
            g = gn
        else:
            g = ops.render_bwd(helios, sun, normals, trig, ctx.trig_stride, plane, rays,
                               xs, ys, c(g_image), c(g_actual), c(g_refl))
            if gn is not None:
                g = g + gn
        return g, None, None, None, None, None, None


def env_step_fused(field, sun, normals, consts: StepConstants, notify: bool = False):
    """One autograd node for HelioEnv.step: → (image, actual, refl [B,N,3], mse, dist, bound,
    alignment_loss, mae [B], angles [B,N], all_bounds [B,N], flag, ticket) — ``ticket`` (an int,
    0 without ``notify``) is the step's completion ticket for ``ops.notify_wait``."""
    trig, stride = field._select_trig(sun.shape[0])
    field._receiver()            # (an in-place write to a receiver tensor since the last call → fresh records)
    node = getattr(_field._get_ops(), "env_step_node", None)
    out = node(field, sun, normals, trig, stride, consts, notify) if node is not None else None
    if out is not None:          # the same node as a C++ autograd Function (csrc/hostbind.cpp)
        return out
    return _EnvStep.apply(normals, field, sun, trig, stride, consts, notify)


def step_losses(img, actual, action, consts: StepConstants):
    """→ (mse, dist, bound, alignment_loss, mae [B], angles [B,N], all_bounds [B,N], flag)."""
    img, actual, action = img.contiguous(), actual.contiguous(), action.contiguous()
    if torch.is_grad_enabled() and (img.requires_grad or actual.requires_grad or action.requires_grad):
        return _StepLosses.apply(img, actual, action, consts)
    out, mae, align, allb, _ = _field._get_ops().step_losses_fwd(img, actual, action, consts)
    return out[0], out[1], out[2], out[3], mae, align, allb, out[4]
