"""Sun-batch sharding of HelioField.render over the GPUs of one node.

Images are independent per sun position (the reference sums only over heliostats,
newenv_rl_test_multi_error.py:404-406), so the batch axis shards with no data-path
collective: rank g renders rows [g·chunk, (g+1)·chunk) of the global batch for ALL
heliostats, with the matching rows of the pre-sampled error tensor, and the only
exchange is one all-gather of the rendered images (RCCL over xGMI with the "nccl"
backend; gloo in the CPU tests).  The backward needs no collective: every rank
consumes the rows of ∂L/∂image it produced.  One process per GPU.
"""
from __future__ import annotations

import torch

from .comm import ImageGather


def shard_rows(B: int, world: int, rank: int):
    """Rows [b0, b1) of a global batch of ``B`` owned by ``rank``: fixed-size chunks of
    ceil(B/world) in rank order (trailing ranks may own fewer rows, or none)."""
    chunk = -(-B // world)
    b0 = min(B, rank * chunk)
    return b0, min(B, b0 + chunk), chunk


class _GatherRows(torch.autograd.Function):
    """all-gather of equally sized row blocks; backward = this rank's block of the cotangent."""

    @staticmethod
    def forward(ctx, local, gather):
        out = torch.empty((gather.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        gather.gather(local, out)
        ctx.rows, ctx.rank = local.shape[0], gather.rank
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows], None


class ShardedRenderer:
    """Renders a global batch sharded over the ranks of ``group`` with one HelioField per rank.

    Every rank passes the SAME global ``sun`` [B,3] and ``action`` [B,3N] (replicated
    host-side state, like the reference's single process) and gets back the full
    ``images`` [B,R,R]; ``actual`` / ``refl`` are returned for the local rows unless
    ``gather_geometry`` is set.
    """

    def __init__(self, field, group=None, transport: str = "auto"):
        self.field = field
        self.group = group
        self.gather = ImageGather(group, transport)     # RCCL directly on GPUs, torch.distributed otherwise
        self.world, self.rank = self.gather.world, self.gather.rank

    def local_rows(self, B: int):
        return shard_rows(B, self.world, self.rank)[:2]

    def render(self, sun_position, action, monitor: bool = False, gather_geometry: bool = False):
        f = self.field
        sun = torch.as_tensor(sun_position, dtype=torch.float32, device=f.device).reshape(-1, 3)
        B, N = sun.shape[0], f.num_heliostats
        act = torch.as_tensor(action, dtype=torch.float32, device=f.device).reshape(B, N, 3)
        b0, b1, chunk = shard_rows(B, self.world, self.rank)

        R = f.resolution
        if b1 > b0:
            out = f.render_rows(sun[b0:b1], act[b0:b1], b0, B, monitor=True)
            img, actual, refl = out
        else:   # more ranks than rows: this rank contributes padding only — empty row blocks that still hang off
            # `action` in the autograd graph, so that a replicated loss differentiates on every rank (to zeros here)
            hook = act[:0].sum(dim=(1, 2))
            img = hook.view(0, 1, 1).expand(0, R, R)
            actual = hook.view(0, 1, 1).expand(0, N, 3)
            refl = hook.view(0, 1).expand(0, 3)
        if self.world == 1:
            images = img
        else:
            if img.shape[0] < chunk:    # pad the ragged tail so every rank sends `chunk` rows
                img = torch.cat([img, img.new_zeros((chunk - img.shape[0], R, R))], dim=0)
            images = _GatherRows.apply(img, self.gather)[:B]
            if gather_geometry:
                pad = lambda t, rows: torch.cat([t, t.new_zeros((rows - t.shape[0],) + tuple(t.shape[1:]))], 0)  # noqa: E731
                actual = _GatherRows.apply(pad(actual, chunk), self.gather)[:B]
                refl = _GatherRows.apply(pad(refl.view(-1, N, 3), chunk), self.gather)[:B].reshape(-1, 3)
        return (images, actual, refl) if monitor else (images, actual)
