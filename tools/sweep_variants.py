#!/usr/bin/env python3
"""Which splat-forward kernel wins where: times variants 3 (regs 128²), 4 (tile 128²), 5 (tile 256²),
6 (regs 64²) over a (B, N, R) grid and prints the automatic choice beside the measured best."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native
from bench import time_kernel

dev = torch.device("cuda")
ops = native.get_ops(); lib = ops.lib
st = native._stream()
print(f"{'B':>5} {'N':>5} {'R':>4} | " + " ".join(f"v{v:>1}(us)".rjust(10) for v in (3, 4, 5, 6)) + " |  auto(us)  best")
for R in (64, 128, 256, 512):
    xs = torch.linspace(-7.5, 7.5, R, device=dev); ys = xs.clone()
    for N in (50, 500, 5000):
        for B in (4, 16, 64, 256):
            if B * N * R * R > 4e11: continue
            rays = torch.rand(B, N, 4, device=dev) * torch.tensor([10., 10., 0.5, 0.01], device=dev) - torch.tensor([5., 5., 0., 0.], device=dev)
            img = torch.empty(B, R, R, device=dev)
            res = {}
            for v in (3, 4, 5, 6, 0):
                args = (B, N, R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), img.data_ptr(), v, None, 0, st)
                flops = 2.0 * B * N * R * R
                iters = max(3, min(100, int(3e11 / flops)))
                res[v] = time_kernel(lambda: lib.helio_splat_fwd(*args), iters, warm=2) * 1e6
            best = min((3, 4, 5, 6), key=lambda v: res[v])
            flag = "" if res[0] <= 1.1 * res[best] else "   <-- auto is >10% off"
            print(f"{B:5d} {N:5d} {R:4d} | " + " ".join(f"{res[v]:10.1f}" for v in (3, 4, 5, 6)) + f" | {res[0]:9.1f}  v{best}{flag}")
