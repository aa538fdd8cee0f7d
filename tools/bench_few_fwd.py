#!/usr/bin/env python3
"""The footprint forward where it IS HBM-bound: a handful of rays per image (the reference's TTT sweeps run ONE
heliostat and 500 suns, run_experiments.py:31-56) — render_fwd_few, one launch for the whole render, bound by writing
the image once (4·B·R² bytes; with HelioEnv.step's loss block in the same launch also 8·B·R² read).  HIP-event time of
helio_render_fwd (variant 13 = the few-ray form) and of the env-step forward, GB/s of algorithmic bytes against 8 TB/s.
usage: bench_few_fwd.py [out.txt]            (the table)
       bench_few_fwd.py B N R                (one shape, for profiler passes: every launch of the run is that kernel at that size)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

dev = torch.device("cuda")
ops = native.get_ops()
lines = []
def emit(s):
    print(s, flush=True); lines.append(s)
emit("# render_fwd_few: one launch = the whole render; algorithmic bytes = 4*B*R*R written (+ 44 B per ray)")
SHAPES = ((512, 1, 512), (512, 2, 512), (512, 4, 512), (512, 8, 512), (500, 1, 128), (4096, 1, 128), (4096, 8, 128), (64, 8, 1024))
if len(sys.argv) == 4:
    SHAPES = ((int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])),)
for B, N, R in SHAPES:
    w = synthetic.Workload("few", N=N, B=B, R=R, sigma_scale=0.02, error_scale_mrad=40.0)
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    act = make_action(f, suns_d, noise)
    trig, stride = f._select_trig(B)
    normals = act.reshape(B, N, 3).contiguous()
    hp, pl, xs, ys = f.heliostat_positions, f._plane, f._xs, f._ys
    with torch.no_grad():
        rays = ops.render_fwd(hp, suns_d, normals, trig, stride, pl, xs, ys)[3]
        nbytes = 4.0 * B * R * R + 44.0 * B * N
        iters = max(20, min(400, int(4e10 / nbytes)))
        assert ops.render_choice(B, N, R) == 13, ops.render_choice(B, N, R)
        t = time_kernel(lambda: ops.render_fwd(hp, suns_d, normals, trig, stride, pl, xs, ys, rays=rays, want_refl=False, variant=13), iters, warm=5, repeats=3)
    emit(f"B={B:5d} N={N} R={R:5d}: render_fwd_few {t * 1e6:8.1f} us = {nbytes / t / 1e9:7.1f} GB/s = {nbytes / t / 8e12:.3f} of 8 TB/s   ({nbytes / 1e6:.1f} MB)")
    del f, rays
    torch.cuda.empty_cache()
if len(sys.argv) == 2:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
