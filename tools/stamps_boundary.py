#!/usr/bin/env python3
"""The boundary between two CONSECUTIVE config-2 launches, from inside the kernels (diagnostic build, tools/build_diag.py).

Every wave of the fused kernel stamps s_memrealtime (100 MHz, one clock for the whole chip) at entry and after its last
store has completed; with `helio_diag_set_slot` two launches enqueued back to back write their stamps to different halves of
the buffer.  From pairs (A, B) of consecutive launches out of a long back-to-back run:
    span(A)    = last wave end of A − first wave start of A            (the kernel as the GPU sees it)
    boundary   = first wave start of B − last wave end of A            (dispatch: end-of-kernel → next kernel's first wave)
    period     = first wave start of B − first wave start of A         (= span + boundary: what a long loop pays per step)
next to the period of the PRODUCT kernel over 2000 launches (HIP events) and to bench.py's figure for a step.
usage: stamps_boundary.py [out.txt]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

w = synthetic.CONFIGS["cfg2"]
dev = torch.device("cuda")
import build_diag
path = build_diag.build()
diag = ctypes.CDLL(path)
vp, i, l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
diag.helio_render_fwd.restype = i
diag.helio_render_fwd.argtypes = [i, i, i, vp, vp, vp, vp, l, ctypes.POINTER(native.Plane), vp, vp, vp, vp, vp, vp, i, vp, l, vp]
diag.helio_diag_set_stamps.restype = i
diag.helio_diag_set_stamps.argtypes = [vp]
diag.helio_diag_set_slot_words.restype = i
diag.helio_diag_set_slot_words.argtypes = [l]
diag.helio_diag_set_slot.restype = None
diag.helio_diag_set_slot.argtypes = [i]
diag.helio_diag_fused_kg.restype = i
diag.helio_diag_fused_kg.argtypes = [i, i, i]

helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev)
suns_d = suns.to(dev)
act = make_action(f, suns_d, noise)
trig, stride = f._select_trig(w.B)
normals = act.reshape(w.B, w.N, 3).contiguous()
actual = torch.empty_like(normals)
rays = torch.empty(w.B, w.N, 4, device=dev)
img = torch.empty(w.B, w.R, w.R, device=dev)
st = native._stream()
args = (w.B, w.N, w.R, f.heliostat_positions.data_ptr(), suns_d.data_ptr(), normals.data_ptr(), trig.data_ptr(), stride,
        f._plane, f._xs.data_ptr(), f._ys.data_ptr(), actual.data_ptr(), None, rays.data_ptr(), img.data_ptr(), 0, None, 0, st)
blocks = ((w.R + 31) // 32) ** 2
NW, NS = diag.helio_diag_fused_kg(w.B, w.N, w.R), 12
words = w.B * blocks * NW * NS
stamps = torch.zeros(2 * words, dtype=torch.int64, device=dev)
assert diag.helio_diag_set_stamps(stamps.data_ptr()) == 0 and diag.helio_diag_set_slot_words(words) == 0

spans, bounds, periods = [], [], []
for trial in range(40):
    # a long back-to-back run, slots alternating: its last two launches are consecutive and both still in the buffer
    for k in range(300):
        diag.helio_diag_set_slot(k & 1)
        diag.helio_render_fwd(*args)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(2, w.B * blocks, NW, NS).astype(np.int64)
    A, Bv = s[0], s[1]                      # launch 298 (slot 0), launch 299 (slot 1)
    a0, a1, b0 = A[:, :, 8].min(), A[:, :, 9].max(), Bv[:, :, 8].min()
    spans.append((a1 - a0) * 10.0)
    bounds.append((b0 - a1) * 10.0)
    periods.append((b0 - a0) * 10.0)
lib = native.get_ops().lib
t_prod = time_kernel(lambda: lib.helio_render_fwd(*args), 2000, warm=200)
t_diag = time_kernel(lambda: diag.helio_render_fwd(*args), 2000, warm=200)
lines = []
def emit(x):
    print(x, flush=True); lines.append(x)
q = lambda v: f"median {np.median(v):7.0f} ns   [p10 {np.percentile(v, 10):7.0f} … p90 {np.percentile(v, 90):7.0f}]"
emit(f"# {w.name}: two consecutive launches of render_fwd_fused_small<{NW},false> (grid {blocks} x {w.B} workgroups), s_memrealtime stamps, 40 pairs out of back-to-back runs of 300")
emit(f"span of a launch (first wave start → last wave's stores complete)   {q(spans)}")
emit(f"boundary (last wave end of A → first wave start of B)              {q(bounds)}")
emit(f"period (first wave start of A → first wave start of B)             {q(periods)}")
emit(f"# HIP-event period over 2000 back-to-back launches: stamped build {t_diag * 1e6:.2f} us, product kernel {t_prod * 1e6:.2f} us")
emit("# (the stamps' fences forbid overlaps the product kernel has: the stamped build's span is an upper bound of the product kernel's;")
emit("#  the boundary is the dispatcher's and does not depend on the build)")
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
