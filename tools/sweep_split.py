#!/usr/bin/env python3
"""Few images of many heliostats: the 256² LDS-table kernel with the heliostat sum split across workgroups
(variants 14..17 = 2, 4, 8, 16 parts, partial images in the caller's scratch) against what the size rule picks
without it (HELIO_SPLIT=0: k-split blocks / 128² register tiles).  One helio_render_fwd call per (B, N, R),
HIP-event time, fraction of the 157.3 TFLOP/s f32 peak by the dense 2·B·N·R² flops.
usage: sweep_split.py [err_mrad] [sigma_scale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

err = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
dev = torch.device("cuda")
ops = native.get_ops()
PEAK = 157.3
variants = (3, 9, 5, 14, 15, 16, 17)
print(f"err = {err} mrad, sigma_scale = {sigma}; us per render (frac of peak); * = the size rule's choice")
print(f"{'B':>4} {'N':>5} {'R':>4} | " + " ".join(f"{('v' + str(v)):>14}" for v in variants))
grid = [(R, N, B) for R in (256, 512) for N in (500, 1000, 2000, 5000) for B in (2, 4, 8, 16, 32, 64, 128)]
for R, N, B in grid:
    if B * N * R * R > 2e11:
        continue
    w = synthetic.Workload("s", N=N, B=B, R=R, sigma_scale=sigma, error_scale_mrad=err, span=30.0)
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev, max_batch=max(B, 2))
    suns_d = suns.to(dev)
    act = make_action(f, suns_d, noise)
    trig, stride = f._select_trig(B)
    normals = act.reshape(B, N, 3).contiguous()
    flops = 2.0 * B * N * R * R
    iters = max(5, min(100, int(1e11 / flops)))
    choice = ops.render_choice(B, N, R)
    cells = []
    with torch.no_grad():
        rays = ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys)[3]
        for v in variants:
            try:
                t = time_kernel(lambda: ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys,
                                                       rays=rays, variant=v), iters, warm=3)
                cells.append(f"{t*1e6:8.1f} ({flops/t/1e12/PEAK:4.2f}){'*' if v == choice else ' '}")
            except RuntimeError:
                cells.append(f"{'-':>14} ")
    print(f"{B:4d} {N:5d} {R:4d} | " + " ".join(cells), flush=True)
    del f, rays
    torch.cuda.empty_cache()
