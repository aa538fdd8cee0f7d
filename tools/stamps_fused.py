#!/usr/bin/env python3
"""Where the config-2 render kernel spends its cycles: s_memtime stamps of the fused small-problem
kernel (diagnostic build, tools/build_diag.py → $TMPDIR/libhelio_diag.so, built on demand).

    python tools/stamps_fused.py [cfg] > profiles/r02_fused_stamps.txt

Per wave of every workgroup: cycles between stamps (median / p90 over the workgroups of the LAST of
200 back-to-back launches), and — from s_memrealtime (100 MHz) — when workgroups start and end
relative to the first start of the grid.  Shares are what to read, not the run time: the stamps'
fences forbid overlaps the real kernel has."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = synthetic.CONFIGS[cfg]
dev = torch.device("cuda")
import build_diag            # the diagnostic library is built on demand, outside the tree (hipcc, ≈30 s)
path = build_diag.OUT if os.path.exists(build_diag.OUT) else build_diag.build()
diag = ctypes.CDLL(path)
vp, i, l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
diag.helio_render_fwd.restype = i
diag.helio_render_fwd.argtypes = [i, i, i, vp, vp, vp, vp, l, ctypes.POINTER(native.Plane), vp, vp, vp, vp, vp, vp, i, vp, l, vp]
diag.helio_diag_set_stamps.restype = i
diag.helio_diag_set_stamps.argtypes = [vp]

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
diag.helio_diag_fused_kg.restype = i
diag.helio_diag_fused_kg.argtypes = [i, i, i]
blocks = ((w.R + 31) // 32) ** 2
NW, NS = diag.helio_diag_fused_kg(w.B, w.N, w.R), 12
stamps = torch.zeros(w.B * blocks * NW * NS, dtype=torch.int64, device=dev)
assert diag.helio_diag_set_stamps(stamps.data_ptr()) == 0
t_diag = time_kernel(lambda: diag.helio_render_fwd(*args), 200, warm=50)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(w.B * blocks, NW, NS).astype(np.int64)
# the product kernel on the same inputs, for the period the shares are to be applied to
lib = native.get_ops().lib
t_prod = time_kernel(lambda: lib.helio_render_fwd(*args), 2000, warm=200)
img_diag = img.clone()
img.zero_()
lib.helio_render_fwd(*args)
torch.cuda.synchronize()
assert torch.equal(img_diag, img), "diagnostic build and product kernel disagree"

phases = [((0, 1), "entry → ray loads landed, trace_head done AND late kernel arguments arrived"),
          ((1, 2), "trace_tail, ray table in LDS, barrier passed"),
          ((2, 3), "heliostat loop (factors + MFMA)"),
          ((3, 4), "exchange of the partial sums (LDS, barrier), sum"),
          ((4, 5), "image stores issued"),
          ((5, 6), "image stores complete")]
print(f"# {w.name}: fused kernel, grid {blocks} x {w.B} workgroups (one 32x32 block each) of {NW} waves; wave 0 traces the rays (N <= 64), the others wait for the late arguments and the barrier")
print(f"# back-to-back period: product kernel {t_prod*1e6:.2f} us, stamped build {t_diag*1e6:.2f} us")
clk = []
for wg in range(s.shape[0]):
    for wv in range(NW):
        dt, dr = s[wg, wv, 6] - s[wg, wv, 0], s[wg, wv, 9] - s[wg, wv, 8]
        if dr > 0:
            clk.append(dt / (dr * 10e-9) / 1e9)
ghz = float(np.median(clk))
print(f"# shader clock inside the kernel: {ghz:.2f} GHz (s_memtime / s_memrealtime)")
print(f"# cycles per phase, median [p10 … p90] over the {s.shape[0]} workgroups")
print(f"{'phase':76s} " + " ".join(f"{'wave %d' % k:>22s}" for k in range(NW)))


def cell(d):
    return f"{np.median(d):7.0f} [{np.percentile(d, 10):5.0f}…{np.percentile(d, 90):5.0f}]" if d.size else f"{'-':>22s}"


for (a, b), name in phases:
    cells = []
    for wv in range(NW):
        ok = (s[:, wv, a] > 0) & (s[:, wv, b] > 0)
        cells.append(cell((s[:, wv, b] - s[:, wv, a])[ok]))
    print(f"{name:76s} " + " ".join(cells))
tot = s[:, :, 6] - s[:, :, 0]
print(f"{'entry → stores complete (whole wave)':76s} " + " ".join(cell(tot[:, k]) for k in range(NW)))
print(f"# = {np.median(tot)/ghz/1e3:.2f} us of the {t_prod*1e6:.2f} us period; the rest is dispatch: launch → first wave, ramp, "
      f"end-of-kernel → next launch")
r0 = s[:, :, 8].min()
starts, ends = (s[:, :, 8] - r0) * 10.0, (s[:, :, 9] - r0) * 10.0          # ns
print(f"# wave start after the first start of the grid (ns): median {np.median(starts):.0f}, p90 {np.percentile(starts, 90):.0f}, "
      f"max {starts.max():.0f}")
print(f"# wave end   after the first start of the grid (ns): median {np.median(ends):.0f}, p90 {np.percentile(ends, 90):.0f}, "
      f"max {ends.max():.0f}   (grid span = {ends.max()/1e3:.2f} us)")
