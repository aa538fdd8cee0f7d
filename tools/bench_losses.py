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
t_b = time_kernel(lambda: ops.step_losses_bwd(img, actual, action, c, one, one, one, one, None, True, True, True), 20)
by_f = 12.0 * B * R * R + 48.0 * B * N          # img, target, dmap + ideal/actual/action (+8 B/ray out)
by_b = 16.0 * B * R * R + 60.0 * B * N          # + grad_img write, grad_actual/grad_action writes
print(f"B={B} N={N} R={R}: step_losses fwd {t_f*1e6:8.1f} us = {by_f/t_f/1e9:7.1f} GB/s ({by_f/t_f/8e12*100:4.1f}% of 8 TB/s) | "
      f"bwd {t_b*1e6:8.1f} us = {by_b/t_b/1e9:7.1f} GB/s ({by_b/t_b/8e12*100:4.1f}%)")

# the few-ray moment kernel (backward of the footprints for a handful of rays per image): grad_image streamed once
Nf = 1
rays = torch.rand(B, Nf, 4, device=dev, generator=g) * torch.tensor([10., 10., 0.5, 0.01], device=dev) - torch.tensor([5., 5., 0., 0.], device=dev)
xs = torch.linspace(-7.5, 7.5, R, device=dev); ys = xs.clone()
G = torch.randn(B, R, R, device=dev, generator=g)
t_few = time_kernel(lambda: ops.splat_bwd(rays, xs, ys, G, variant=4), 20)
print(f"B={B} N={Nf} R={R}: splat_bwd_few (grad_image read once)   {t_few*1e6:8.1f} us = {4.0*B*R*R/t_few/1e9:7.1f} GB/s ({4.0*B*R*R/t_few/8e12*100:4.1f}%)")
# the same with the image cotangent of mse/dist formed on the fly (helio_env_step_bwd): img, target, distance map read once
from doodle_amd import native as _n
lib = ops.lib
mom = torch.empty(B, lib.helio_splat_bwd_blocks(R), Nf, 5, device=dev)
grad = torch.empty(B, Nf, 3, device=dev)
plane = _n.Plane(); plane.origin[:] = (0, -5, 0); plane.normal[:] = (0, 1, 0); plane.u[:] = (1, 0, 0); plane.v[:] = (0, 0, 1); plane.w[:] = (0, -1, 0); plane.sigma_scale = 0.01
sun = torch.rand(B, 3, device=dev, generator=g) * 1e4
act1 = unit(torch.rand(B, Nf, 3, device=dev, generator=g)); trig = torch.tensor([1., 0., 1., 0.], device=dev).repeat(B, Nf, 1).contiguous()
ideal1 = unit(torch.rand(B, Nf, 3, device=dev, generator=g)); h1 = helios[:Nf].contiguous()
def fused():
    _n._check(lib, lib.helio_env_step_bwd(B, Nf, R, h1.data_ptr(), sun.data_ptr(), act1.data_ptr(), trig.data_ptr(), 4 * Nf, plane,
                                         rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), img.data_ptr(), target.data_ptr(), c.tx.data_ptr(),
                                         dm.data_ptr(), ideal1.data_ptr(), c.tp, c.tn, 15.0, 15.0, 0, None, one.data_ptr(), None, None, None,
                                         None, None, None, mom.data_ptr(), grad.data_ptr(), 0, _n._stream()))
t_fused = time_kernel(fused, 20)
print(f"B={B} N={Nf} R={R}: helio_env_step_bwd (few-ray, fused loss adjoint + geometry backward) {t_fused*1e6:8.1f} us = {12.0*B*R*R/t_fused/1e9:7.1f} GB/s ({12.0*B*R*R/t_fused/8e12*100:4.1f}%)")
