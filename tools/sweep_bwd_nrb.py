#!/usr/bin/env python3
"""Tuning run for bwd_small_nrb(): the small-tile backward (variant 3) with HELIO_BWD_NRB = 1 / 2 / 4 ray
blocks per wave (read once per process), per shape; variant 2 beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native
from bench import time_kernel

dev = torch.device("cuda")
ops = native.get_ops(); lib = ops.lib; st = native._stream()
shapes = [(25, 1000, 128), (25, 1000, 256), (25, 300, 128), (25, 5000, 128), (25, 200, 256), (8, 2000, 256), (100, 300, 128),
          (25, 100, 128), (64, 1000, 64), (4, 5000, 256), (25, 600, 512), (256, 1000, 64)]
if len(sys.argv) > 1:
    print("shapes:", " ".join(f"({b},{n},{r})" for b, n, r in shapes))
out3, out2 = [], []
for B, N, R in shapes:
    xs = torch.linspace(-7.5, 7.5, R, device=dev); ys = xs.clone()
    rays = torch.rand(B, N, 4, device=dev) * torch.tensor([10., 10., 0.5, 0.01], device=dev) - torch.tensor([5., 5., 0., 0.], device=dev)
    G = torch.randn(B, R, R, device=dev)
    mom = torch.empty(B, lib.helio_splat_bwd_blocks(R), N, native.MOMENT_STRIDE, device=dev)
    for v, out in ((3, out3), (2, out2)):
        args = (B, N, R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), G.data_ptr(), mom.data_ptr(), v, None, 0, st)
        iters = max(5, min(100, int(2e11 / (4.0 * B * N * R * R))))
        out.append(f"{min(time_kernel(lambda: lib.helio_splat_bwd(*args), iters, warm=2) for _ in range(2)) * 1e6:7.1f}")
print(f"NRB={os.environ.get('HELIO_BWD_NRB', 'rule'):5s} v3 " + " ".join(out3), flush=True)
if len(sys.argv) > 1:
    print(f"          v2 " + " ".join(out2), flush=True)
