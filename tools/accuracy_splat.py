#!/usr/bin/env python3
"""Accuracy of the splat-forward kernels against an fp64 evaluation of the same footprints
(same f32 ray parameters and pixel coordinates, factors and sums in double).
usage: accuracy_splat.py [cfg] [B] variants..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
w = synthetic.CONFIGS[cfg]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
variants = [int(v) for v in sys.argv[3:]] or [1, 5, 7]
w = synthetic.Workload(w.name, w.N, B, w.R, w.sigma_scale, w.error_scale_mrad, w.span)
dev = torch.device("cuda")
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev)
suns_d = suns.to(dev)
act = make_action(f, suns_d, noise)
ops = native.get_ops()
trig, stride = f._select_trig(B)
_, _, rays = ops.geometry_fwd(f.heliostat_positions, suns_d, act.reshape(B, w.N, 3).contiguous(), trig, stride, f._plane)
r = rays.double()
xs, ys = f._xs.double(), f._ys.double()
truth = torch.empty(B, w.R, w.R, dtype=torch.float64, device=dev)
for b in range(B):
    a, bb, k2, c2 = (r[b, :, i:i + 1] for i in range(4))
    A = torch.exp2(-((xs[None, :] + a) ** 2 + c2) * k2)          # [N,R]
    E = torch.exp2(-((ys[None, :] + bb) ** 2) * k2)
    truth[b] = A.t() @ E
peak = truth.max().item()
sig = truth > 1e-6 * peak
print(f"{w.name} B={B}: fp64 truth, peak {peak:.4f}, {sig.float().mean().item()*100:.1f}% of pixels above 1e-6 of peak")
for v in variants:
    img = ops.splat_fwd(rays, f._xs, f._ys, variant=v).double()
    rel = ((img - truth) / truth)[sig]
    print(f"variant {v}: max|d|/peak {((img-truth).abs().max()/peak).item():.2e} | per-pixel relative error: "
          f"max {rel.abs().max().item():.2e}, mean signed {rel.mean().item():+.2e}, rms {rel.pow(2).mean().sqrt().item():.2e}")
