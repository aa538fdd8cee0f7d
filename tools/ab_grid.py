#!/usr/bin/env python3
"""render forward / backward (helio_render_fwd / helio_render_bwd as the size rules choose the kernels) over the
B x N x R grid of tools/sweep_render.py with the library of a given source tree — for A/B runs of two trees, or of
one tree with HELIO_CULL=0 / 1, on ONE box.  Least of five timing loops per figure.

    AB_ERR=90 AB_SIGMA=0.01 [HELIO_CULL=0] python tools/ab_grid.py <tree root>        (default: err 40 mrad, sigma 0.02)

profiles/r03_g_grid_ab.txt: round-2 tree vs this one, and lists on / off, at err 40 / 90 / 180."""
import os, sys
root = os.path.abspath(sys.argv[1]); sys.path.insert(0, root)
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action
dev = torch.device("cuda"); ops = native.get_ops()
def best(fn, iters, reps=5):
    for _ in range(5): fn()
    out = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        out = min(out, e0.elapsed_time(e1) * 1e3 / iters)
    return out
shapes = [(B, N, R) for R in (64, 128, 256, 512) for N in (50, 200, 1000, 5000) for B in (4, 32, 256) if B * N * R * R <= 3e11]
ERR, SIG = float(os.environ.get("AB_ERR", "40")), float(os.environ.get("AB_SIGMA", "0.02"))
for (B, N, R) in shapes:
    w = synthetic.Workload("s", N=N, B=B, R=R, sigma_scale=SIG, error_scale_mrad=ERR, span=30.0 if N > 100 else 10.0)
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
    trig, stride = f._select_trig(B); normals = act.reshape(B, N, 3).contiguous()
    G = torch.randn(B, R, R, device=dev)
    with torch.no_grad():
        out = ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys); rays = out[3]
        it = max(5, min(200, int(1e11 / (2.0 * B * N * R * R))))
        tf = best(lambda: ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys, rays=rays), it)
        tb = best(lambda: ops.render_bwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, rays, f._xs, f._ys, G, None, None), it)
    del out
    print(f"{os.path.basename(root):10s} B={B:4d} N={N:5d} R={R:4d} fwd {tf:8.1f} us  bwd {tb:8.1f} us", flush=True)
