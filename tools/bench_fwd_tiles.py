#!/usr/bin/env python3
"""The forward footprint kernels that take lists — 128² register tiles (3), 128² table tiles (4), 256² table tiles (5) — dense and with
their lists, at the bench's large configurations and a few grid sizes, err / sigma of the configuration (default: the reference's 90 mrad /
0.01): smaller tiles cull more (a ray is dropped per TILE), larger ones run closer to the peak.  helio_splat_fwd per call (HIP events,
least of three loops).   usage: bench_fwd_tiles.py [B N R span] ..."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

dev = torch.device("cuda")
ops = native.get_ops()
cfg4 = synthetic.CONFIGS["cfg4"]
cfg5 = dataclasses.replace(synthetic.CONFIGS["cfg5"], B=512)
sizes = [("cfg4", cfg4), ("cfg5 shard", cfg5)]
for B, N, R, span in ((256, 5000, 256, 30.0), (256, 1000, 256, 30.0), (32, 1000, 512, 30.0), (32, 5000, 512, 30.0), (256, 200, 256, 30.0), (256, 1000, 512, 30.0)):
    sizes.append((f"B={B} N={N} R={R} span={span:g}", synthetic.Workload("s", N=N, B=B, R=R, span=span)))
SIGMA, ERR = float(os.environ.get("HELIO_SIGMA", 0.01)), float(os.environ.get("HELIO_ERR", 90.0))      # (of the sizes given on the command line)
if len(sys.argv) > 1:        # B N R span quadruples instead
    a = sys.argv[1:]
    sizes = [(f"B={a[i]} N={a[i + 1]} R={a[i + 2]} span={a[i + 3]}", synthetic.Workload("s", N=int(a[i + 1]), B=int(a[i]), R=int(a[i + 2]), span=float(a[i + 3]),
                                                                                sigma_scale=SIGMA, error_scale_mrad=ERR))
             for i in range(0, len(a) - 3, 4)]
print(f"{'size':28s} | " + " | ".join(f"v{v} dense   lists  live" for v in (3, 4, 5)) + " | auto = v, us")
for name, w in sizes:
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    act = make_action(f, suns_d, noise)
    trig, stride = f._select_trig(w.B)
    normals = act.reshape(w.B, w.N, 3).contiguous()
    cells = []
    with torch.no_grad():
        rays = ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys)[3]
        xs, ys = f._xs, f._ys
        ref = None
        for v in (3, 4, 5):
            iters = 10 if w.B * w.N * w.R * w.R > 1e11 else 30
            dense = time_kernel(lambda: ops.splat_fwd(rays, xs, ys, variant=v, cull=False), iters, warm=2, repeats=3)
            lists = time_kernel(lambda: ops.splat_fwd(rays, xs, ys, variant=v), iters, warm=2, repeats=3)
            n = ops.lib.helio_fwd_scratch_bytes(w.B, w.N, w.R, v)
            live = float("nan")
            if n:
                te = 256 if v == 5 else 128
                t = -(-w.R // te)
                scratch = torch.zeros(n, dtype=torch.uint8, device=dev)
                img = torch.empty(w.B, w.R, w.R, device=dev)
                ops.lib.helio_splat_fwd(w.B, w.N, w.R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), img.data_ptr(), v, scratch.data_ptr(), n, native._stream())
                torch.cuda.synchronize()
                live = scratch[:4 * w.B * t * t].view(torch.int32).float().mean().item() / w.N
            cells.append(f"{dense * 1e6:8.1f} {lists * 1e6:7.1f} {live:5.3f}")
        auto = time_kernel(lambda: ops.splat_fwd(rays, xs, ys, variant=0), iters, warm=2, repeats=3)
    print(f"{name:28s} | " + " | ".join(cells) + f" | v{ops.lib.helio_render_fwd_choice(w.B, w.N, w.R)} {auto * 1e6:.1f}", flush=True)
    del f, rays
    torch.cuda.empty_cache()
