#!/usr/bin/env python3
"""Experiment (round 4): the backward for images of at most 128 pixels across in 128 c x 128 rays tiles (4 waves, 34 KB of LDS,
several workgroups per CU; HELIO_BWD_WR2=1, dense launches only) against the default 128 c x 256 rays (8 waves, one
workgroup per CU), the 64-ray tiles (variant 12) and the default with lists: helio_splat_bwd per call (HIP events, least of
three loops)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

dev = torch.device("cuda")
ops = native.get_ops()
print(f"{'B':>4} {'N':>5} {'R':>4} | {'256 rays':>9} {'128 rays':>9} {'64 rays':>9} | {'256+lists':>9} | of the f32 MFMA peak (best dense) | same bits")
SIZES = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]       # B,N,R … instead of the table's sizes
for B, N, R in SIZES or ((500, 50, 128), (500, 200, 128), (256, 200, 128), (500, 200, 100), (32, 1000, 128), (256, 1000, 128), (4, 5000, 128),
                (32, 5000, 128), (256, 5000, 128), (256, 1000, 64), (256, 5000, 64)):
    w = synthetic.Workload("s", N=N, B=B, R=R, sigma_scale=0.02, error_scale_mrad=40.0)
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    act = make_action(f, suns_d, noise)
    trig, stride = f._select_trig(B)
    _, _, rays = ops.geometry_fwd(f.heliostat_positions, suns_d, act.reshape(B, N, 3).contiguous(), trig, stride, f._plane)
    G = torch.randn(B, R, R, device=dev)
    xs, ys = f._xs, f._ys
    res, out = {}, {}
    for key, mode, variant, cull in (("256", "0", 2, False), ("128", "1", 2, False), ("64", "0", 12, False), ("lists", "0", 2, None)):
        os.environ["HELIO_BWD_WR2"] = mode
        out[key] = ops.splat_bwd(rays, xs, ys, G, variant=variant, cull=cull)
        torch.cuda.synchronize()
        res[key] = time_kernel(lambda: ops.splat_bwd(rays, xs, ys, G, variant=variant, cull=cull), 20, warm=3, repeats=3)
    os.environ["HELIO_BWD_WR2"] = "0"
    same = all(torch.equal(out["256"].view(torch.int32), out[k].view(torch.int32)) for k in ("128", "64", "lists"))
    best = min(res["256"], res["128"], res["64"])
    print(f"{B:4d} {N:5d} {R:4d} | {res['256'] * 1e6:9.1f} {res['128'] * 1e6:9.1f} {res['64'] * 1e6:9.1f} | {res['lists'] * 1e6:9.1f} | "
          f"{4.0 * B * N * R * R / best / 157.3e12:.3f} | {same}", flush=True)
    del f, G, rays, out
    torch.cuda.empty_cache()
