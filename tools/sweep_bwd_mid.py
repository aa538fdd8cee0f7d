#!/usr/bin/env python3
"""Where the LDS-tile backward (variant 2, both passes in one launch) overtakes the small-tile kernel (variant 3):
helio_splat_bwd dense, both variants, at sizes around the rule of splat_bwd_choice (`tiles` = workgroups of ONE pass
of variant 2).  Least of three timing loops."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native
from bench import time_kernel

dev = torch.device("cuda")
ops = native.get_ops(); lib = ops.lib; st = native._stream()
print(f"{'B':>5} {'N':>5} {'R':>4} {'tiles':>6} | {'v2 (us)':>9} {'v3 (us)':>9} | auto (us)  chosen")
shapes = []
for R in (128, 256, 512):
    for N in (300, 1000, 5000):
        per_image = (-(-R // (128 if R <= 128 else 256))) * (-(-N // 256))
        for tiles in (16, 32, 64, 96, 128, 192):
            B = max(1, tiles // per_image)
            if (B, N, R) not in [s[:3] for s in shapes] and B * N * R * R <= 2e10:
                shapes.append((B, N, R, B * per_image))
for B, N, R, tiles in shapes:
    xs = torch.linspace(-7.5, 7.5, R, device=dev); ys = xs.clone()
    rays = torch.rand(B, N, 4, device=dev) * torch.tensor([10., 10., 0.5, 0.01], device=dev) - torch.tensor([5., 5., 0., 0.], device=dev)
    G = torch.randn(B, R, R, device=dev)
    mom = torch.empty(B, lib.helio_splat_bwd_blocks(R), N, native.MOMENT_STRIDE, device=dev)
    res = {}
    for v in (2, 3, 0):
        args = (B, N, R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), G.data_ptr(), mom.data_ptr(), v, None, 0, st)
        iters = max(5, min(200, int(1e11 / (4.0 * B * N * R * R))))
        res[v] = time_kernel(lambda: lib.helio_splat_bwd(*args), iters, warm=3, repeats=3) * 1e6
    chosen = 2 if abs(res[0] - res[2]) < abs(res[0] - res[3]) else 3
    flag = "" if res[chosen] <= 1.05 * min(res[2], res[3]) else "   <-- the other is faster"
    print(f"{B:5d} {N:5d} {R:4d} {tiles:6d} | {res[2]:9.1f} {res[3]:9.1f} | {res[0]:9.1f}  v{chosen}{flag}", flush=True)
