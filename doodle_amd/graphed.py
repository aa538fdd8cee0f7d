"""HIP-graph replay of the launch-bound inner loops around HelioEnv.step.

The reference's test-time-compute loop (train_with_env_com_trunc_advantage_ttt.py:291-312) calls
``env.step(candidate)`` and ``loss.backward()`` hundreds of times per optimiser step on problems of
a few microseconds of GPU work each: five kernels (render + loss partials, finishing workgroup,
loss backward, splat backward, geometry backward) behind ≈100 µs of Python, autograd-engine and
launch overhead.  Every kernel of this package enqueues on the caller's stream, allocates through
torch's caching allocator and never synchronises, so the whole iteration can be captured once
with ``torch.cuda.graph`` and replayed as a single graph launch.

``GraphedEnvStep`` does that for the common shape of the loop: a static input tensor, an optional
differentiable ``prepare`` (e.g. ``normalize(base + fine_error_vec)``), ``env.step`` and the
gradient of one metric.  Anything else (an optimiser with ``capturable=True``, several steps per
graph) can be captured the same way by the caller — nothing here is special-cased.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


class GraphedEnvStep:
    """``env.step(prepare(x))`` and ``d metrics[objective] / d x`` as one HIP graph.

    >>> g = GraphedEnvStep(env, like=fine_error_vec, prepare=lambda v: F.normalize(base + v, dim=2))
    >>> metrics, grad = g(fine_error_vec)          # replays; tensors are overwritten by the next call

    The graph bakes in the env's current sun positions, errors and distance maps (call
    ``recapture()`` after ``set_sun_pos`` / ``reset_errors``).  The reference's NaN/Inf asserts
    (test_environment.py:495-501) cannot run inside a graph; ``nonfinite()`` reads the flag the
    finishing workgroup wrote (one device→host read, only when asked for).
    """

    def __init__(self, env, like: torch.Tensor, objective: str = "dist",
                 prepare: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, warmup: int = 3):
        if not like.is_cuda:
            raise RuntimeError("GraphedEnvStep needs a HIP device tensor; there is no CPU path")
        self.env, self.objective, self.prepare, self.warmup = env, objective, prepare, warmup
        self.x = like.detach().clone().requires_grad_(True)        # the graph's static input
        self.recapture()

    def _iteration(self):
        action = self.x if self.prepare is None else self.prepare(self.x)
        obs, metrics, monitor = self.env.step(action)
        (grad,) = torch.autograd.grad(metrics[self.objective], self.x)
        return obs, metrics, monitor, grad

    def recapture(self):
        env = self.env
        saved, env.check_finite = env.check_finite, False          # a host wait cannot be captured
        try:
            env._reference()                                       # cached constants: not part of the graph
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self._iteration()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.obs, self.metrics, self.monitor, self.grad = self._iteration()
        finally:
            env.check_finite = saved

    def __call__(self, x: Optional[torch.Tensor] = None):
        """Replay on ``x`` (copied into the static input; None: the caller updated ``self.x`` in
        place).  → (metrics, grad) — static tensors, valid until the next call."""
        if x is not None:
            with torch.no_grad():
                self.x.copy_(x)
        self.graph.replay()
        return self.metrics, self.grad

    def nonfinite(self) -> bool:
        """The reference's NaN/Inf asserts for the last replay (synchronises)."""
        vals = torch.stack([self.metrics["mse"], self.metrics["dist"], self.metrics["bound"]])
        return not bool(torch.isfinite(vals).all())


class GraphedRenderGrad:
    """``HelioField.render`` + a caller-supplied scalar loss of ``(image, actual)`` + its gradient
    w.r.t. the action as one HIP graph (BASELINE config 3: forward + backward through the render).

    >>> g = GraphedRenderGrad(field, sun, like=action, loss=lambda img, actual: (img * G).sum() + actual.sum())
    >>> loss, grad = g(action)            # replays; the tensors are overwritten by the next call

    Eagerly that iteration is the render's autograd node, the caller's few torch ops and PyTorch's
    autograd engine — 85–210 µs of host time for ≈30 µs of kernels at config 3; replayed, it is one
    graph launch.  The graph bakes in the field's current errors and ``sigma_scale`` (``recapture()``
    after ``reset_errors()``) and the sun positions passed here.  For a loss whose cotangents are
    known in closed form, :meth:`HelioField.render_value_and_grad` needs neither graph nor autograd.
    """

    def __init__(self, field, sun_position, like: torch.Tensor, loss: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                 warmup: int = 3):
        if not like.is_cuda:
            raise RuntimeError("GraphedRenderGrad needs a HIP device tensor; there is no CPU path")
        self.field, self.loss_fn, self.warmup = field, loss, warmup
        self.sun = torch.as_tensor(sun_position, dtype=torch.float32, device=like.device).clone()
        self.x = like.detach().clone().requires_grad_(True)        # the graph's static input
        self.recapture()

    def _iteration(self):
        img, actual = self.field.render(self.sun, self.x, None)
        loss = self.loss_fn(img, actual)
        (grad,) = torch.autograd.grad(loss, self.x)
        return img, actual, loss, grad

    def recapture(self):
        self.field._select_trig(self.sun.shape[0] if self.sun.dim() > 1 else 1)   # trig table: not part of the graph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._iteration()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.image, self.actual, self.loss, self.grad = self._iteration()

    def __call__(self, x: Optional[torch.Tensor] = None):
        """Replay on ``x`` (copied into the static input; None: the caller updated ``self.x`` in place).
        → (loss, grad) — static tensors, valid until the next call; ``self.image`` / ``self.actual`` too."""
        if x is not None:
            with torch.no_grad():
                self.x.copy_(x)
        self.graph.replay()
        return self.loss, self.grad
