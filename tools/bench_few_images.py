#!/usr/bin/env python3
"""Few images of a large field — one or a few suns over a whole plant — forward and backward as the size rules choose
the kernels (k-split blocks / small-tile whole-k form), with the lists of csrc/cull.h (default) and dense
(HELIO_CULL=0 in the environment).  usage: bench_few_images.py [err_mrad] [sigma_scale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

err = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
dev = torch.device("cuda"); ops = native.get_ops()
print(f"err = {err} mrad, sigma_scale = {sigma}, HELIO_CULL = {os.environ.get('HELIO_CULL', '1')}")
for B, N, R in ((1, 5000, 512), (2, 5000, 512), (4, 5000, 512), (8, 2000, 512), (4, 5000, 256), (16, 5000, 256), (25, 1000, 128), (25, 3000, 256)):
    w = synthetic.Workload("s", N=N, B=B, R=R, sigma_scale=sigma, error_scale_mrad=err, span=30.0)
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev, max_batch=max(B, 2)); sd = suns.to(dev); act = make_action(f, sd, noise)
    trig, stride = f._select_trig(B); normals = act.reshape(B, N, 3).contiguous()
    G = torch.randn(B, R, R, device=dev)
    with torch.no_grad():
        rays = ops.render_fwd(f.heliostat_positions, sd, normals, trig, stride, f._plane, f._xs, f._ys)[3]
        tf = time_kernel(lambda: ops.render_fwd(f.heliostat_positions, sd, normals, trig, stride, f._plane, f._xs, f._ys, rays=rays), 30, repeats=3)
        tb = time_kernel(lambda: ops.render_bwd(f.heliostat_positions, sd, normals, trig, stride, f._plane, rays, f._xs, f._ys, G, None, None), 30, repeats=3)
    print(f"B={B:3d} N={N:5d} R={R:4d}: fwd {tf*1e6:7.1f} us (variant {ops.render_choice(B, N, R):2d}, scratch {ops.lib.helio_fwd_scratch_bytes(B, N, R, 0):9d} B)   "
          f"bwd {tb*1e6:7.1f} us (scratch {ops.lib.helio_bwd_scratch_bytes(B, N, R, 0):8d} B)", flush=True)
