#!/usr/bin/env python3
"""HBM roofline of the fused step-loss kernels (the genuinely HBM-bound piece of the path)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native
from doodle_amd.losses import StepConstants
from bench import time_kernel

B, N, R = (int(x) for x in (sys.argv[1:4] or (512, 2000, 512)))
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
img, target, dm = (torch.rand(B, R, R, device=dev, generator=g) for _ in range(3))
unit = lambda t: t / t.norm(dim=-1, keepdim=True)
ideal = unit(torch.rand(B, N, 3, device=dev, generator=g)); actual = unit(ideal + 0.01 * torch.rand(B, N, 3, device=dev, generator=g))
action = unit(ideal + 0.1 * torch.rand(B, N, 3, device=dev, generator=g))
helios = torch.rand(N, 3, device=dev, generator=g) * 10 + 80
f3 = ctypes.c_float * 3
c = StepConstants(target, target.amax((1, 2)).clamp_min(1e-6), dm, ideal, helios, f3(0, -5, 0), f3(0, 1, 0), 15.0, 15.0, False)
ops = native.get_ops()
t_f = time_kernel(lambda: ops.step_losses_fwd(img, actual, action, c), 20)
one = torch.ones((), device=dev)
t_b = time_kernel(lambda: ops.step_losses_bwd(img, actual, action, c, one, one, one, one, True, True, True), 20)
by_f = 12.0 * B * R * R + 48.0 * B * N          # img, target, dmap + ideal/actual/action (+8 B/ray out)
by_b = 16.0 * B * R * R + 60.0 * B * N          # + grad_img write, grad_actual/grad_action writes
print(f"B={B} N={N} R={R}: step_losses fwd {t_f*1e6:8.1f} us = {by_f/t_f/1e9:7.1f} GB/s ({by_f/t_f/8e12*100:4.1f}% of 8 TB/s) | "
      f"bwd {t_b*1e6:8.1f} us = {by_b/t_b/1e9:7.1f} GB/s ({by_b/t_b/8e12*100:4.1f}%)")
