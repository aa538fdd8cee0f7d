#!/usr/bin/env python3
"""One shape under rocprofv3: 200 x (helio_render_fwd, helio_render_bwd) as the size rules choose the kernels.
usage: profile_shape.py B N R"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action

B, N, R = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda")
ops = native.get_ops()
w = synthetic.Workload("s", N=N, B=B, R=R, sigma_scale=0.02, error_scale_mrad=40.0, span=30.0 if N > 100 else 10.0)
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev)
suns_d = suns.to(dev)
act = make_action(f, suns_d, noise)
trig, stride = f._select_trig(B)
normals = act.reshape(B, N, 3).contiguous()
G = torch.randn(B, R, R, device=dev)
with torch.no_grad():
    out = ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys)
    rays = out[3]
    for _ in range(200):
        ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys, rays=rays)
        ops.render_bwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, rays, f._xs, f._ys, G, None, None)
torch.cuda.synchronize()
print("choice", ops.render_choice(B, N, R))
