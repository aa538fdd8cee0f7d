#!/usr/bin/env python3
"""A/B helper (round 4): the forward LDS-table kernel at the sizes where its last 64-ray chunk matters — dense and with the
lists — one line per size: helio_splat_fwd per call (HIP events, least of five loops).  Run once per build of libhelio.so
(tools/ab_fwd_last.sh swaps the library between runs); argv[1] = a label for the lines."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

label = sys.argv[1] if len(sys.argv) > 1 else "-"
dev = torch.device("cuda")
ops = native.get_ops()
cfg4 = synthetic.CONFIGS["cfg4"]
cfg5 = dataclasses.replace(synthetic.CONFIGS["cfg5"], B=512)
SIZES = [("cfg4", cfg4, 5, 10), ("cfg5 shard", cfg5, 5, 10),
         ("B=256 N=200 R=512", synthetic.Workload("s", N=200, B=256, R=512, sigma_scale=0.02, error_scale_mrad=40.0), 5, 30),
         ("B=256 N=50 R=512", synthetic.Workload("s", N=50, B=256, R=512, sigma_scale=0.02, error_scale_mrad=40.0), 5, 30),
         ("B=500 N=200 R=256", synthetic.Workload("s", N=200, B=500, R=256, sigma_scale=0.02, error_scale_mrad=40.0), 5, 30),
         ("B=500 N=330 R=256", synthetic.Workload("s", N=330, B=500, R=256, sigma_scale=0.02, error_scale_mrad=40.0), 5, 30)]
for name, w, variant, iters in SIZES:
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    act = make_action(f, suns_d, noise)
    trig, stride = f._select_trig(w.B)
    normals = act.reshape(w.B, w.N, 3).contiguous()
    with torch.no_grad():
        rays = ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys)[3]
        xs, ys = f._xs, f._ys
        dense = time_kernel(lambda: ops.splat_fwd(rays, xs, ys, variant=variant, cull=False), iters, warm=3, repeats=5)
        lists = time_kernel(lambda: ops.splat_fwd(rays, xs, ys, variant=variant), iters, warm=3, repeats=5)
        a = ops.splat_fwd(rays, xs, ys, variant=variant, cull=False)
        b = ops.splat_fwd(rays, xs, ys, variant=variant)
        same = torch.equal(a.view(torch.int32), b.view(torch.int32))
        csum = a.double().sum().item()
    fl = 2.0 * w.B * w.N * w.R * w.R / 1e12
    print(f"{label:8s} {name:20s} v{variant}: dense {dense * 1e6:9.1f} us = {fl / dense / 157.3:6.4f} of 157.3 TFLOP/s | with lists {lists * 1e6:9.1f} us | "
          f"lists == dense bits: {same} | sum {csum:.9e}", flush=True)
    del f, rays, a, b
    torch.cuda.empty_cache()
