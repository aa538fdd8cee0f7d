"""Synthetic heliostat fields for benchmarks and tests.

All tensors are drawn with a CPU ``torch.Generator`` so that the CPU baseline and every
GPU rank see identical inputs; geometry follows the reference's training script
(train_with_env.py:227-231): heliostats ``rand(N,3)·span+80`` with z=0, target (0,−5,0),
normal (0,1,0), 15×15 m receiver, suns on the upper hemisphere at radius hypot(1e4,1e4).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

SUN_RADIUS = math.hypot(1e4, 1e4)


@dataclass
class Workload:
    name: str
    N: int
    B: int
    R: int
    sigma_scale: float = 0.01
    error_scale_mrad: float = 90.0
    span: float = 10.0


# BASELINE.json "configs", in order
CONFIGS = {
    "cfg1": Workload("N=50,B=1,R=128", 50, 1, 128),
    "cfg2": Workload("N=50,B=25,R=128", 50, 25, 128),
    "cfg4": Workload("N=2000,B=512,R=512", 2000, 512, 512, span=60.0),
    "cfg5": Workload("N=5000,B=4096,R=256", 5000, 4096, 256, span=100.0),
}


def make_inputs(w: Workload, seed: int = 0, b_offset: int = 0, b_count: int | None = None):
    """helios [N,3], suns [b_count,3], errors [b_count,N,2] (mrad), noise [b_count,N,3] — CPU fp32.

    Rows ``b_offset : b_offset+b_count`` of the global batch: a rank that renders a shard
    draws exactly the rows it owns (each row has its own generator seed), so sharded and
    unsharded runs see the same per-sun inputs.
    """
    g = torch.Generator().manual_seed(seed)
    helios = torch.rand(w.N, 3, generator=g) * w.span + 80.0
    helios[:, 2] = 0.0
    b_count = w.B if b_count is None else b_count
    suns, errs, noise = [], [], []
    for b in range(b_offset, b_offset + b_count):
        gb = torch.Generator().manual_seed(1_000_003 * (seed + 1) + b)
        d = torch.randn(3, generator=gb)
        d = d / d.norm()
        d[2] = d[2].abs()
        suns.append(d * SUN_RADIUS)
        errs.append(torch.randn(w.N, 2, generator=gb) * w.error_scale_mrad)
        noise.append(torch.randn(w.N, 3, generator=gb) * 0.01)
    return helios, torch.stack(suns).float(), torch.stack(errs).float(), torch.stack(noise).float()


TARGET_POSITION = (0.0, -5.0, 0.0)
TARGET_NORMAL = (0.0, 1.0, 0.0)
TARGET_AREA = (15.0, 15.0)
