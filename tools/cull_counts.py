#!/usr/bin/env python3
"""How many rays the culling stage keeps (doodle_amd/csrc/cull.h) at a BASELINE config: per (image, tile) for the
forward lists, per image for the backward list.   usage: cull_counts.py [cfg] [B]      (HELIO_ERR / HELIO_SIGMA)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
w = synthetic.CONFIGS[cfg]
B = int(sys.argv[2]) if len(sys.argv) > 2 else min(w.B, 512)
w = synthetic.Workload(w.name, w.N, B, w.R, float(os.environ.get("HELIO_SIGMA", w.sigma_scale)),
                       float(os.environ.get("HELIO_ERR", w.error_scale_mrad)), w.span)
dev = torch.device("cuda")
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
ops = native.get_ops(); lib = ops.lib
trig, stride = f._select_trig(B)
_, _, rays = ops.geometry_fwd(f.heliostat_positions, suns_d, act.reshape(B, w.N, 3).contiguous(), trig, stride, f._plane)
N, R = w.N, w.R
img = torch.empty(B, R, R, device=dev)
for variant, te in ((5, 256), (3, 128)):
    nb = lib.helio_fwd_scratch_bytes(B, N, R, variant)
    if not nb:
        continue
    s = torch.zeros(nb, dtype=torch.uint8, device=dev)
    assert lib.helio_splat_fwd(B, N, R, rays.data_ptr(), f._xs.data_ptr(), f._ys.data_ptr(), img.data_ptr(), variant,
                               s.data_ptr(), nb, native._stream()) == 0
    t = -(-R // te)
    c = s[:4 * B * t * t].view(torch.int32).float()
    print(f"forward, {te}x{te} tiles: live fraction {c.mean().item() / N:.3f} (min {int(c.min())}, max {int(c.max())} of {N}); "
          f"in 64-ray chunks {(torch.ceil(c / 64) * 64).mean().item() / (-(-N // 64) * 64):.3f}")
G = torch.randn(B, R, R, device=dev)
mom = torch.empty(B, lib.helio_splat_bwd_blocks(R), N, 5, device=dev)
nb = lib.helio_bwd_scratch_bytes(B, N, R, 2)
if nb:
    s = torch.zeros(nb, dtype=torch.uint8, device=dev)
    assert lib.helio_splat_bwd(B, N, R, rays.data_ptr(), f._xs.data_ptr(), f._ys.data_ptr(), G.data_ptr(), mom.data_ptr(), 2,
                               s.data_ptr(), nb, native._stream()) == 0
    ct = -(-R // 256)
    T = B * ct * 2 if 2 <= ct <= 8 else B             # one list per (pass, image, c tile) where an image is 2..8 tiles wide
    c = s[:4 * T].view(torch.int32).float()
    print(f"backward, {T // B} list(s) per image: live fraction {c.mean().item() / N:.3f} (min {int(c.min())}, max {int(c.max())} of {N}); "
          f"in 256-ray tiles {(torch.ceil(c / 256)).mean().item() / (-(-N // 256)):.3f}")
