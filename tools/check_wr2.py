#!/usr/bin/env python3
"""Experiment: the large backward entirely in the 128-ray form of the LDS-tile kernel (two 8-wave workgroups per CU,
HELIO_BWD_WR2=1, dense launches only) against the 256-ray form: same bits?  how long?   usage: check_wr2.py [cfg] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
w0 = synthetic.CONFIGS[cfg]
B = int(sys.argv[2]) if len(sys.argv) > 2 else w0.B
w = synthetic.Workload(w0.name, w0.N, B, w0.R, w0.sigma_scale, w0.error_scale_mrad, w0.span)
dev = torch.device("cuda")
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
ops = native.get_ops()
trig, stride = f._select_trig(w.B)
_, _, rays = ops.geometry_fwd(f.heliostat_positions, suns_d, act.reshape(w.B, w.N, 3).contiguous(), trig, stride, f._plane)
G = torch.randn(w.B, w.R, w.R, device=dev)
fl = 4.0 * w.B * w.N * w.R * w.R
out = {}
for mode in ("0", "1"):
    os.environ["HELIO_BWD_WR2"] = mode
    m = ops.splat_bwd(rays, f._xs, f._ys, G, variant=2, cull=False)
    torch.cuda.synchronize()
    t = time_kernel(lambda: ops.splat_bwd(rays, f._xs, f._ys, G, variant=2, cull=False), 10, warm=2, repeats=2)
    out[mode] = m
    print(f"{w.name} B={B} dense, HELIO_BWD_WR2={mode}: {t * 1e6:9.1f} us = {fl / t / 1e12 / 157.3:.3f} of the f32 MFMA peak", flush=True)
print("same bits:", torch.equal(out["0"].view(torch.int32), out["1"].view(torch.int32)))
