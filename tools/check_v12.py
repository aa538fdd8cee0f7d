#!/usr/bin/env python3
"""Experiment (round 4): backward variant 12 — the LDS-tile kernel in 64-ray tiles (4 waves, 16 k per chunk) — against the
rules' choice, for fields of 33–256 heliostats on large or many images: helio_render_bwd per call (HIP events, least of three
loops of 20).  Same bits as variant 2 (256-ray tiles)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

dev = torch.device("cuda")
ops = native.get_ops()
WIDE = len(sys.argv) > 1 and sys.argv[1] == "wide"        # fields just past one or two 256-ray tiles; column v8 is then variant 9
SIGMA, ERR = (float(sys.argv[-2]), float(sys.argv[-1])) if len(sys.argv) > 2 else (0.02, 40.0)     # e.g. 0.01 90: the reference's defaults, where the lists bite
print(f"# sigma_scale {SIGMA}, error_scale_mrad {ERR}")
print(f"{'B':>4} {'N':>4} {'R':>4} | {'auto':>9} {'=v':>3} | {'v12':>9} {'v2':>9} {'v10':>9} {'v8':>9} | {'v12 dense':>9} {'v2 dense':>9} | v12 == v2 bits")
OTHER = 9 if WIDE else 8
for R in ((128, 256, 512) if WIDE else (100, 128, 256, 512)):
    for N in ((260, 300, 320, 384, 448, 520, 576, 640) if WIDE else (96, 128, 160, 200) if len(sys.argv) > 1 and sys.argv[1] == "small" else (40, 50, 64, 96, 128, 200) if R <= 128 else (33, 160, 192)):
        for B in ((32, 128, 500) if WIDE else (500, 2048) if len(sys.argv) > 1 and sys.argv[1] == "small" else (60, 256, 500)):
            if B * N * R * R > 2e10:
                continue
            w = synthetic.Workload("s", N=N, B=B, R=R, sigma_scale=SIGMA, error_scale_mrad=ERR, span=30.0 if N > 100 else 10.0)
            helios, suns, errs, noise = synthetic.make_inputs(w, 0)
            f = build_field(w, helios, errs, dev)
            suns_d = suns.to(dev)
            act = make_action(f, suns_d, noise)
            trig, stride = f._select_trig(B)
            normals = act.reshape(B, N, 3).contiguous()
            hp, pl, xs, ys = f.heliostat_positions, f._plane, f._xs, f._ys
            G = torch.randn(B, R, R, device=dev)
            with torch.no_grad():
                rays = ops.render_fwd(hp, suns_d, normals, trig, stride, pl, xs, ys)[3]
                res = {}
                for v in (0, 12, 2, 10, OTHER):
                    try:
                        ops.render_bwd(hp, suns_d, normals, trig, stride, pl, rays, xs, ys, G, None, None, variant=v)
                        torch.cuda.synchronize()
                        res[v] = time_kernel(lambda: ops.render_bwd(hp, suns_d, normals, trig, stride, pl, rays, xs, ys, G, None, None, variant=v), 20, warm=3, repeats=3)
                    except RuntimeError:
                        res[v] = float("nan")
                ops.cull = False                      # no scratch: the dense kernels
                for v in (12, 2):
                    res[f"{v}d"] = time_kernel(lambda: ops.render_bwd(hp, suns_d, normals, trig, stride, pl, rays, xs, ys, G, None, None, variant=v), 20, warm=3, repeats=3)
                ops.cull = True
                same = torch.equal(ops.splat_bwd(rays, xs, ys, G, variant=12, cull=False).view(torch.int32),
                                   ops.splat_bwd(rays, xs, ys, G, variant=2, cull=False).view(torch.int32))
            print(f"{B:4d} {N:4d} {R:4d} | {res[0] * 1e6:9.1f} {ops.render_bwd_choice(B, N, R):3d} | {res[12] * 1e6:9.1f} {res[2] * 1e6:9.1f} {res[10] * 1e6:9.1f} {res[OTHER] * 1e6:9.1f} | {res['12d'] * 1e6:9.1f} {res['2d'] * 1e6:9.1f} | {same}", flush=True)
            del f, G, rays
            torch.cuda.empty_cache()
