#!/usr/bin/env python3
"""Tuning run for ksplit_parts(): one process per HELIO_KSPLIT = 4 / 8 / 16 waves a block (read once per
process); prints splat variant 9 forced to that form, per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native
from bench import time_kernel

dev = torch.device("cuda")
ops = native.get_ops(); lib = ops.lib; st = native._stream()
tag = f"kp={os.environ.get('HELIO_KSPLIT', 'rule')}"
out = []
for B, N, R in [(25, 1000, 128), (25, 1000, 256), (25, 5000, 256), (40, 1000, 256), (64, 1000, 256), (25, 2000, 192), (4, 5000, 512),
                (16, 1000, 512), (100, 1000, 128), (100, 5000, 128), (64, 600, 256), (25, 300, 256)]:
    xs = torch.linspace(-7.5, 7.5, R, device=dev); ys = xs.clone()
    rays = torch.rand(B, N, 4, device=dev) * torch.tensor([10., 10., 0.5, 0.01], device=dev) - torch.tensor([5., 5., 0., 0.], device=dev)
    img = torch.empty(B, R, R, device=dev)
    args = (B, N, R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), img.data_ptr(), 9, None, 0, st)
    iters = max(5, min(100, int(3e11 / (2.0 * B * N * R * R))))
    t = min(time_kernel(lambda: lib.helio_splat_fwd(*args), iters, warm=2) for _ in range(2)) * 1e6
    out.append(f"{t:7.1f}")
print(f"{tag:16s} " + " ".join(out), flush=True)
