#!/usr/bin/env python3
"""Which splat-backward kernel wins where (1 VALU, 2 MFMA 256-tiles, 3 MFMA 64-tiles, 4 few-ray streaming)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native
from bench import time_kernel

dev = torch.device("cuda")
ops = native.get_ops(); lib = ops.lib
st = native._stream()
print(f"{'B':>5} {'N':>5} {'R':>4} | " + " ".join(f"v{v}(us)".rjust(10) for v in (1, 2, 3, 4)) + " |  auto(us)  best")
for R in (64, 128, 256, 512):
    xs = torch.linspace(-7.5, 7.5, R, device=dev); ys = xs.clone()
    jb = lib.helio_splat_bwd_blocks(R)
    for N in (1, 2, 4, 8, 16, 32, 50, 500, 5000):
        for B in (4, 25, 64, 256, 500):
            if B * N * R * R > 2e11: continue
            rays = torch.rand(B, N, 4, device=dev) * torch.tensor([10., 10., 0.5, 0.01], device=dev) - torch.tensor([5., 5., 0., 0.], device=dev)
            G = torch.randn(B, R, R, device=dev)
            mom = torch.empty(B, jb, N, 5, device=dev)
            res = {}
            for v in (1, 2, 3, 4, 0):
                if v == 4 and N > 64:
                    res[v] = float("inf")
                    continue
                args = (B, N, R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), G.data_ptr(), mom.data_ptr(), v, None, 0, st)
                iters = max(3, min(100, int(1e11 / (2.0 * B * N * R * R))))
                res[v] = time_kernel(lambda: lib.helio_splat_bwd(*args), iters, warm=2) * 1e6
            best = min((1, 2, 3, 4), key=lambda v: res[v])
            flag = "" if res[0] <= 1.1 * res[best] else "   <-- auto is >10% off"
            print(f"{B:5d} {N:5d} {R:4d} | " + " ".join(f"{res[v]:10.1f}" for v in (1, 2, 3, 4)) + f" | {res[0]:9.1f}  v{best}{flag}")
