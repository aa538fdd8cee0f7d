#!/usr/bin/env python3
"""Efficiency map of the render's forward and backward as the size rules choose the kernels: one
helio_render_fwd / helio_render_bwd call per (B, N, R), HIP-event time per call, TFLOP/s by the
2·B·N·R² (forward) / 4·B·N·R² (backward) MFMA flops of the separable form, and the fraction of the
157.3 TFLOP/s f32 peak.  Each figure is the least of three timing loops: on the shared boxes one loop in
fifty is stalled for 10–80 ms by something outside the process (seen at different sizes from run to run,
never in 240 loops of a dedicated probe), which a single mean shows as a 10× outlier.
usage: sweep_render.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

dev = torch.device("cuda")
ops = native.get_ops()
PEAK = 157.3
print(f"{'B':>5} {'N':>5} {'R':>4} | {'fwd us':>9} {'TF':>6} {'frac':>5} | {'bwd us':>9} {'TF':>6} {'frac':>5} | launches fwd/bwd-fused")
grid = [(R, N, B) for R in (64, 128, 256, 512) for N in (50, 200, 1000, 5000) for B in (4, 32, 256)]
for R, N, B in grid:
    if B * N * R * R > 3e11:
        continue
    w = synthetic.Workload("s", N=N, B=B, R=R, sigma_scale=0.02, error_scale_mrad=40.0, span=30.0 if N > 100 else 10.0)
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    act = make_action(f, suns_d, noise)
    trig, stride = f._select_trig(B)
    normals = act.reshape(B, N, 3).contiguous()
    G = torch.randn(B, R, R, device=dev)
    flops = 2.0 * B * N * R * R
    iters = max(5, min(200, int(2e11 / flops)))
    with torch.no_grad():
        out = ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys)
        rays = out[3]
        t_f = time_kernel(lambda: ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys, rays=rays), iters, warm=3, repeats=3)
        t_b = time_kernel(lambda: ops.render_bwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, rays, f._xs, f._ys, G, None, None), iters, warm=3, repeats=3)
    lf = ops.lib.helio_render_fwd_launches(B, N, R)
    print(f"{B:5d} {N:5d} {R:4d} | {t_f*1e6:9.1f} {flops/t_f/1e12:6.1f} {flops/t_f/1e12/PEAK:5.2f} | {t_b*1e6:9.1f} {2*flops/t_b/1e12:6.1f} {2*flops/t_b/1e12/PEAK:5.2f} | {lf}", flush=True)
    del f, G, rays, out
    torch.cuda.empty_cache()
