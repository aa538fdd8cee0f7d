#!/usr/bin/env python3
"""Where the k-split block kernel (splat variant 9) stops paying: helio_splat_fwd per variant over shapes
around the block-count threshold of ksplit_parts().  Run with HELIO_KSPLIT_MAX_BLOCKS=1000000 so that variant 0
shows what the rule WOULD choose with no upper bound."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native
from bench import time_kernel

dev = torch.device("cuda")
ops = native.get_ops(); lib = ops.lib; st = native._stream()
print(f"{'B':>5} {'N':>5} {'R':>4} {'blocks':>7} | " + " ".join(f"v{v}(us)".rjust(10) for v in (3, 5, 6, 9)) + " | best")
for B, N, R in [(64, 1000, 128), (128, 1000, 128), (256, 1000, 64), (512, 1000, 64), (32, 5000, 256), (64, 5000, 256),
                (16, 5000, 512), (8, 1000, 512), (32, 300, 256), (64, 300, 128), (128, 300, 128), (16, 2000, 256),
                (32, 2000, 128), (64, 2000, 128), (24, 500, 256)]:
    xs = torch.linspace(-7.5, 7.5, R, device=dev); ys = xs.clone()
    rays = torch.rand(B, N, 4, device=dev) * torch.tensor([10., 10., 0.5, 0.01], device=dev) - torch.tensor([5., 5., 0., 0.], device=dev)
    img = torch.empty(B, R, R, device=dev)
    res = {}
    for v in (3, 5, 6, 9):
        args = (B, N, R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), img.data_ptr(), v, None, 0, st)
        iters = max(5, min(100, int(3e11 / (2.0 * B * N * R * R))))
        res[v] = min(time_kernel(lambda: lib.helio_splat_fwd(*args), iters, warm=2) for _ in range(2)) * 1e6
    best = min(res, key=res.get)
    print(f"{B:5d} {N:5d} {R:4d} {B * ((R + 31) // 32) ** 2:7d} | " + " ".join(f"{res[v]:10.1f}" for v in (3, 5, 6, 9)) + f" | v{best}", flush=True)
