#!/usr/bin/env python3
"""Backward moment kernels at R <= 128 with many rays: the small-tile kernel (3: 4 / 8 waves) against the
256-wide-tile kernels (2), whose tiles are half padding at R = 128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native
from bench import time_kernel

dev = torch.device("cuda")
ops = native.get_ops(); lib = ops.lib; st = native._stream()
print(f"{'B':>5} {'N':>5} {'R':>4} | " + " ".join(f"v{v}(us)".rjust(10) for v in (2, 3, 6, 7)) + " | auto(us) TF(auto) frac")
for B, N, R in [(256, 5000, 128), (256, 1000, 128), (64, 5000, 128), (256, 5000, 64), (256, 1000, 64), (1024, 1000, 128),
                (256, 300, 128), (64, 1000, 128), (512, 2000, 128), (256, 1000, 192), (256, 1000, 256)]:
    xs = torch.linspace(-7.5, 7.5, R, device=dev); ys = xs.clone()
    rays = torch.rand(B, N, 4, device=dev) * torch.tensor([10., 10., 0.5, 0.01], device=dev) - torch.tensor([5., 5., 0., 0.], device=dev)
    G = torch.randn(B, R, R, device=dev)
    JB = lib.helio_splat_bwd_blocks(R)
    mom = torch.empty(B, JB, N, native.MOMENT_STRIDE, device=dev)
    res = {}
    for v in (2, 3, 6, 7, 0):
        args = (B, N, R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), G.data_ptr(), mom.data_ptr(), v, st)
        iters = max(3, min(50, int(2e11 / (4.0 * B * N * R * R))))
        res[v] = min(time_kernel(lambda: lib.helio_splat_bwd(*args), iters, warm=2) for _ in range(2)) * 1e6
    fl = 4.0 * B * N * R * R
    print(f"{B:5d} {N:5d} {R:4d} | " + " ".join(f"{res[v]:10.1f}" for v in (2, 3, 6, 7)) + f" | {res[0]:8.1f} {fl/res[0]/1e6:7.1f} {fl/res[0]/1e6/157.3:5.2f}", flush=True)
