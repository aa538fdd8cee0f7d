#!/usr/bin/env python3
"""Back-to-back period of the fused small-problem render (C ABI alone, HIP events) at the launch-bound
shapes; run under HELIO_FUSED_KG=1|2|4 to compare the k-split forms (the variable is read once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

dev = torch.device("cuda")
ops = native.get_ops(); lib = ops.lib; st = native._stream()
out = []
for name, N, B, R in (("cfg2", 50, 25, 128), ("cfg1", 50, 1, 128), ("ttt", 1, 500, 128), ("n8", 8, 64, 64), ("n200", 200, 40, 128), ("n256", 256, 8, 256)):
    w = synthetic.Workload(name, N, B, R)
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev, max_batch=max(B, 2)); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
    trig, stride = f._select_trig(B)
    normals = act.reshape(B, N, 3).contiguous()
    actual = torch.empty_like(normals); rays = torch.empty(B, N, 4, device=dev); img = torch.empty(B, R, R, device=dev)
    args = (B, N, R, f.heliostat_positions.data_ptr(), suns_d.data_ptr(), normals.data_ptr(), trig.data_ptr(), stride, f._plane,
            f._xs.data_ptr(), f._ys.data_ptr(), actual.data_ptr(), None, rays.data_ptr(), img.data_ptr(), 0, None, 0, st)
    assert lib.helio_render_fwd_launches(B, N, R) == 1
    t = min(time_kernel(lambda: lib.helio_render_fwd(*args), 3000, warm=300) for _ in range(3))
    out.append(f"{name}(N={N},B={B},R={R}) {t*1e6:6.2f} us")
print(f"HELIO_FUSED_KG={os.environ.get('HELIO_FUSED_KG', 'auto')}: " + " | ".join(out))
