#!/usr/bin/env python3
"""Where do the sporadic stalls of a timing loop come from: host (a late submit) or device (a slow launch)?
Per-call HIP events and host clocks over 30 loops of 200 calls at four shapes; prints every loop whose mean is 1.5x its
median.  (Round 3: none in 240 loops on a quiet box, while a sweep on another box had five 10-80 ms stalls in fifty rows —
something outside the process; the size sweeps therefore take the least of three loops.)"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action
dev = torch.device("cuda")
ops = native.get_ops()
for (B, N, R) in ((4, 5000, 64), (32, 1000, 128), (32, 1000, 64), (4, 1000, 256)):
    w = synthetic.Workload("s", N=N, B=B, R=R, sigma_scale=0.02, error_scale_mrad=40.0, span=30.0)
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
    trig, stride = f._select_trig(B); normals = act.reshape(B, N, 3).contiguous()
    G = torch.randn(B, R, R, device=dev)
    with torch.no_grad():
        out = ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys)
        rays = out[3]
        fns = {"fwd": lambda: ops.render_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, f._xs, f._ys, rays=rays),
               "bwd": lambda: ops.render_bwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, rays, f._xs, f._ys, G, None, None)}
        for name, fn in fns.items():
            for rep in range(30):
                for _ in range(3): fn()
                K = 200
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
                host = []
                torch.cuda.synchronize()
                ev[0].record()
                for i in range(K):
                    t0 = time.perf_counter(); fn(); host.append(time.perf_counter() - t0)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                dt = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
                tot = sum(dt); med = sorted(dt)[K // 2]
                worst = max(range(K), key=lambda i: dt[i])
                if tot > 1.5 * med * K:
                    print(f"B={B} N={N} R={R} {name} rep {rep}: mean {tot/K:.1f} us median {med:.1f}; worst gap #{worst}: device {dt[worst]:.0f} us, host call {host[worst]*1e6:.0f} us; "
                          f"host max {max(host)*1e6:.0f} us at #{max(range(K), key=lambda i: host[i])}; gaps>1ms: {[ (i, round(dt[i])) for i in range(K) if dt[i] > 1000][:6]}", flush=True)
            print(f"B={B} N={N} R={R} {name}: done, last median {med:.1f} us", flush=True)
    del f, G, rays, out
    torch.cuda.empty_cache()
