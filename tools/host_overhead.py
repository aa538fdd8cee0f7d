#!/usr/bin/env python3
"""Where the host time of HelioField.render goes at config 2 (host-bound)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action

def t(fn, n=20000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    el = time.perf_counter() - t0
    torch.cuda.synchronize()
    return el / n * 1e6

w = synthetic.CONFIGS["cfg2"]
dev = torch.device("cuda")
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev)
s = suns.to(dev); a = make_action(f, s, noise)
ops = native.get_ops(); lib = ops.lib
B, N, R = w.B, w.N, w.R
normals = a.reshape(B, N, 3).contiguous()
trig, stride = f._select_trig(B)
actual = torch.empty_like(normals); rays = torch.empty(B, N, 4, device=dev); img = torch.empty(B, R, R, device=dev)
st = torch.cuda.current_stream().cuda_stream
print("render()                 %.2f us" % t(lambda: f.render(s, a, None), 5000))
with torch.no_grad():
    print("render() no_grad         %.2f us" % t(lambda: f.render(s, a, None), 5000))
print("as_tensor(sun)           %.2f" % t(lambda: torch.as_tensor(s, dtype=torch.float32, device=f.device)))
print("reshape+contiguous       %.2f" % t(lambda: a.reshape(B, N, 3).contiguous()))
print("_select_trig             %.2f" % t(lambda: f._select_trig(B)))
print("torch.empty_like         %.2f" % t(lambda: torch.empty_like(normals)))
print("torch.empty((B,R,R))     %.2f" % t(lambda: torch.empty((B, R, R), dtype=torch.float32, device=dev)))
print("current_stream()         %.2f" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("raw stream               %.2f" % t(lambda: torch._C._cuda_getCurrentRawStream(0)))
print("data_ptr                 %.2f" % t(lambda: normals.data_ptr()))
print("is_grad_enabled+req      %.2f" % t(lambda: torch.is_grad_enabled() and normals.requires_grad))
ga = (B, N, f.heliostat_positions.data_ptr(), s.data_ptr(), normals.data_ptr(), trig.data_ptr(), stride, f._plane, actual.data_ptr(), None, rays.data_ptr(), st)
print("ctypes geometry_fwd      %.2f" % t(lambda: lib.helio_geometry_fwd(*ga), 5000))
sa = (B, N, R, rays.data_ptr(), f._xs.data_ptr(), f._ys.data_ptr(), img.data_ptr(), 0, None, 0, st)
print("ctypes splat_fwd         %.2f" % t(lambda: lib.helio_splat_fwd(*sa), 5000))
def both():
    lib.helio_geometry_fwd(*ga); lib.helio_splat_fwd(*sa)
print("ctypes both              %.2f" % t(both, 5000))
print("ops.geometry_fwd         %.2f" % t(lambda: ops.geometry_fwd(f.heliostat_positions, s, normals, trig, stride, f._plane, want_refl=False), 5000))
print("ops.splat_fwd            %.2f" % t(lambda: ops.splat_fwd(rays, f._xs, f._ys), 5000))
