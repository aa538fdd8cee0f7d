#!/usr/bin/env python3
"""The back-to-back period of the smallest kernel of the library on one stream, beside the period of
the config-2 render: how much of the 8.3-8.8 us per render is the dispatch floor of the chip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

ops = native.get_ops(); lib = ops.lib; dev = torch.device("cuda"); st = native._stream()
h = torch.rand(1, 3, device=dev); s = torch.rand(1, 3, device=dev) * 1e4; out = torch.empty(1, 1, 3, device=dev)
tgt = (native.ctypes.c_float * 3)(0.0, -5.0, 0.0)
t_min = time_kernel(lambda: lib.helio_ideal_normals(1, 1, h.data_ptr(), s.data_ptr(), tgt, out.data_ptr(), st), 5000, warm=200)
print(f"smallest kernel (1 thread of work), back-to-back on one stream: {t_min*1e6:.2f} us per launch")
w = synthetic.CONFIGS["cfg2"]
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
trig, stride = f._select_trig(w.B)
normals = act.reshape(w.B, w.N, 3).contiguous()
actual = torch.empty_like(normals); rays = torch.empty(w.B, w.N, 4, device=dev); img = torch.empty(w.B, w.R, w.R, device=dev)
args = (w.B, w.N, w.R, f.heliostat_positions.data_ptr(), suns_d.data_ptr(), normals.data_ptr(), trig.data_ptr(), stride, f._plane,
        f._xs.data_ptr(), f._ys.data_ptr(), actual.data_ptr(), None, rays.data_ptr(), img.data_ptr(), 0, None, 0, st)
t_r = time_kernel(lambda: lib.helio_render_fwd(*args), 5000, warm=200)
print(f"config-2 fused render through the C ABI alone, back-to-back: {t_r*1e6:.2f} us per launch")

# host side of the same call, layer by layer (wall clock per call, GPU kept busy: whichever is slower shows)
import time
def wall(fn, n=20000):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: fn()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best
hb = ops.hb
if hb is not None:
    ph = native._plane_handle(hb, f._plane)
    with torch.no_grad():
        print(f"hb.render_fwd (allocates 4 outputs, launches)            {wall(lambda: hb.render_fwd(ph, f.heliostat_positions, suns_d, normals, trig, stride, f._xs, f._ys, rays, False, 0)):.2f} us")
        print(f"hb.render_any (+ dtype/shape fix-ups)                     {wall(lambda: hb.render_any(ph, f.heliostat_positions, suns_d, act, trig, stride, f._xs, f._ys, rays, False, 0)):.2f} us")
        print(f"ops.render_nograd                                         {wall(lambda: ops.render_nograd(f, suns_d, act, trig, stride, False)):.2f} us")
        print(f"field.render(sun, action, None)                           {wall(lambda: f.render(suns_d, act, None)):.2f} us")
        e = torch.empty
        print(f"torch.empty((25,128,128)) alone                           {wall(lambda: e((25, 128, 128), device=dev)):.2f} us")
        ctx, _, _ = f._render_context(w.B)
        if ctx is not None and hasattr(ctx, "host_costs"):
            for rep in range(2):
                d = ctx.host_costs(suns_d, normals, 20000)
                print("RenderCtx pieces, ns per repetition inside C++:", {k: round(v) for k, v in d.items()})
            print(f"ctx.render(sun, action, False) from Python                {wall(lambda: ctx.render(suns_d, normals, False)):.2f} us")
