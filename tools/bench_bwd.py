#!/usr/bin/env python3
"""Timing of the backward kernels (splat_bwd, geometry_bwd) and of render fwd+bwd.
usage: bench_bwd.py [cfg] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action, time_kernel

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    w = synthetic.CONFIGS[cfg]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else w.B
    w = synthetic.Workload(w.name, w.N, B, w.R, w.sigma_scale, w.error_scale_mrad, w.span)
    dev = torch.device("cuda")
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    act = make_action(f, suns_d, noise)
    ops = native.get_ops()
    trig, stride = f._select_trig(B)
    normals = act.reshape(B, w.N, 3).contiguous()
    actual, refl, rays = ops.geometry_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane)
    G = torch.randn(B, w.R, w.R, device=dev)
    H = torch.randn(B, w.N, 3, device=dev)
    iters = 200 if B * w.N * w.R * w.R < 1e10 else 5
    t_g = time_kernel(lambda: ops.geometry_fwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane), iters)
    t_s = time_kernel(lambda: ops.splat_fwd(rays, f._xs, f._ys), iters)
    t_sb = time_kernel(lambda: ops.splat_bwd(rays, f._xs, f._ys, G, variant=1), iters)
    t_sm = time_kernel(lambda: ops.splat_bwd(rays, f._xs, f._ys, G, variant=2), iters)
    m1 = ops.splat_bwd(rays, f._xs, f._ys, G, variant=1).sum(1)
    m2 = ops.splat_bwd(rays, f._xs, f._ys, G, variant=2).sum(1)
    print("   moments valu vs mfma: max|d|/max =", [(m1[..., k] - m2[..., k]).abs().max().item() / m1[..., k].abs().max().item() for k in range(5)])
    print(f"   splat_bwd mfma {t_sm*1e6:9.1f} us ({2*2.0*B*w.N*w.R*w.R/t_sm/1e12:6.1f} TF of 2 FMA/eval)")
    t_s3 = time_kernel(lambda: ops.splat_bwd(rays, f._xs, f._ys, G, variant=3), iters)
    m3 = ops.splat_bwd(rays, f._xs, f._ys, G, variant=3).sum(1)
    print(f"   splat_bwd mfma-small {t_s3*1e6:9.1f} us; vs valu max|d|/max =", max((m1[..., k] - m3[..., k]).abs().max().item() / m1[..., k].abs().max().item() for k in range(5)))
    if w.R >= 128:
        t_s5 = time_kernel(lambda: ops.splat_bwd(rays, f._xs, f._ys, G, variant=5), iters)
        m5 = ops.splat_bwd(rays, f._xs, f._ys, G, variant=5).sum(1)
        print(f"   splat_bwd split-bf16 (opt-in, variant 5) {t_s5*1e6:9.1f} us ({2*2.0*B*w.N*w.R*w.R/t_s5/1e12:6.1f} TF f32-equivalent); "
              f"vs valu max|d|/max =", max((m1[..., k] - m5[..., k]).abs().max().item() / m1[..., k].abs().max().item() for k in range(5)),
              "| mfma vs valu:", max((m1[..., k] - m2[..., k]).abs().max().item() / m1[..., k].abs().max().item() for k in range(5)))
    mom = ops.splat_bwd(rays, f._xs, f._ys, G)
    t_gb = time_kernel(lambda: ops.geometry_bwd(f.heliostat_positions, suns_d, normals, trig, stride, f._plane, mom, H, None), iters)
    fl = 2.0 * B * w.N * w.R * w.R
    print(f"{w.name} B={B}: geometry_fwd {t_g*1e6:9.1f} us | splat_fwd {t_s*1e6:9.1f} us ({fl/t_s/1e12:6.1f} TF) | "
          f"splat_bwd {t_sb*1e6:9.1f} us ({3*fl/t_sb/1e12:6.1f} TF of 3 FMA/eval) | geometry_bwd {t_gb*1e6:9.1f} us")
    a = act.clone().requires_grad_(True)
    def fb():
        img, actual = f.render(suns_d, a, None)
        torch.autograd.grad((img * G).sum() + (actual * H).sum(), a)
    for _ in range(5): fb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200 if iters > 100 else 5
    for _ in range(n): fb()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n
    with torch.no_grad():
        for _ in range(5): f.render(suns_d, act, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f.render(suns_d, act, None)
        torch.cuda.synchronize(); elf = (time.perf_counter() - t0) / n
    print(f"   render fwd {elf*1e6:9.1f} us/call = {B/elf:12.0f} frames/s | fwd+bwd (autograd) {el*1e6:9.1f} us/call = {B/el:12.0f} frames/s")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: f.render_value_and_grad(suns_d, act, G, H)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n2 = 2000 if iters > 100 else 5
    for _ in range(n2): f.render_value_and_grad(suns_d, act, G, H)
    torch.cuda.synchronize(); elv = (time.perf_counter() - t0) / n2
    print(f"   render_value_and_grad (fwd + bwd kernels back to back, one binding call) {elv*1e6:9.1f} us/call = {B/elv:12.0f} frames/s")

if __name__ == "__main__":
    main()
