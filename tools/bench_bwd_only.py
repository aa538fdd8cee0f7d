#!/usr/bin/env python3
"""Only the large-field splat backward (both passes), for PMC passes: usage bench_bwd_only.py [cfg] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = synthetic.CONFIGS[cfg]
dev = torch.device("cuda")
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
ops = native.get_ops()
trig, stride = f._select_trig(w.B)
normals = act.reshape(w.B, w.N, 3).contiguous()
_, _, rays = ops.geometry_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane)
G = torch.randn(w.B, w.R, w.R, device=dev)
lib = ops.lib
mom = torch.empty(w.B, lib.helio_splat_bwd_blocks(w.R), w.N, 5, device=dev)
mode = os.environ.get("CULL", "1")          # 1: the lists the size query asks for; image: one list per image only; 0: dense
nb = lib.helio_bwd_scratch_bytes(w.B, w.N, w.R, 2) if mode != "0" else 0
if mode == "image":
    pad = lambda n: (n + 255) // 256 * 256      # noqa: E731
    nb = min(nb, pad(4 * w.B) + pad(4 * w.B * w.N) + 256 + 8 * w.B * ((w.N + 255) // 256) + 8 * w.B)
scratch = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
args = (w.B, w.N, w.R, rays.data_ptr(), f._xs.data_ptr(), f._ys.data_ptr(), G.data_ptr(), mom.data_ptr(), 2,
        scratch.data_ptr() if nb else None, nb, native._stream())
t = time_kernel(lambda: lib.helio_splat_bwd(*args), iters)
fl = 2 * 2.0 * w.B * w.N * w.R * w.R
print(f"{w.name} ({'dense' if not nb else 'lists per image' if mode == 'image' else 'culled'}): splat_bwd_mfma (two passes) {t*1e6:.1f} us = {fl/t/1e12:.1f} TFLOP/s = {fl/t/1e12/157.3:.3f} of the f32 MFMA peak")
