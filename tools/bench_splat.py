#!/usr/bin/env python3
"""A/B timing of splat-forward kernel variants on one MI355X (interleaved rounds in one
process, HIP events on the launch stream).  usage: bench_splat.py [cfg] [B] variants...
A variant with a trailing "c" (5c) is run WITH the device scratch: the rays that are exactly zero on a tile
are skipped (doodle_amd/csrc/cull.h); the flop rate printed is always that of the dense work.
HELIO_ERR / HELIO_SIGMA override the workload's error scale (mrad) and sigma_scale."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import native, synthetic
from bench import build_field, make_action

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    w = synthetic.CONFIGS[cfg]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else w.B
    variants = sys.argv[3:] or ["2", "1"]
    w = synthetic.Workload(w.name, w.N, B, w.R, float(os.environ.get("HELIO_SIGMA", w.sigma_scale)),
                           float(os.environ.get("HELIO_ERR", w.error_scale_mrad)), w.span)
    dev = torch.device("cuda")
    helios, suns, errs, noise = synthetic.make_inputs(w, 0)
    f = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    act = make_action(f, suns_d, noise)
    ops = native.get_ops()
    trig, stride = f._select_trig(B)
    _, _, rays = ops.geometry_fwd(f.heliostat_positions, suns_d, act.reshape(B, w.N, 3).contiguous(), trig, stride, f._plane)
    # (PMC passes average over every launch of a kernel: HELIO_NOREF=1 leaves out this dense reference launch)
    ref = None if os.environ.get("HELIO_NOREF") == "1" else ops.splat_fwd(rays, f._xs, f._ys, variant=2, cull=False)
    image = torch.empty((B, w.R, w.R), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * B * w.N * w.R * w.R
    iters = max(3, min(200, int(2e12 / flops)))
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            vi = int(v.rstrip("c"))
            nb = ops.lib.helio_fwd_scratch_bytes(B, w.N, w.R, vi) if v.endswith("c") else 0
            scratch = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            args = (B, w.N, w.R, rays.data_ptr(), f._xs.data_ptr(), f._ys.data_ptr(), image.data_ptr(), vi,
                    scratch.data_ptr() if nb else None, nb, st)
            rc = ops.lib.helio_splat_fwd(*args)
            assert rc == 0, ops.lib.helio_last_error_string()
            torch.cuda.synchronize()
            if rnd == 0 and ref is not None:
                err = (image - ref).abs().max().item() / ref.max().item()
                print(f"variant {v}: max|d|/peak vs variant 2 = {err:.2e}" + (f"  bit-identical: {torch.equal(image, ref)}" if vi == 5 else ""))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.lib.helio_splat_fwd(*args)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) * 1e-3 / iters)
    for v in variants:
        ts = sorted(res[v])
        print(f"variant {v:>4s}: median {ts[len(ts)//2]*1e6:9.1f} us  min {ts[0]*1e6:9.1f} us  "
              f"{flops/ts[len(ts)//2]/1e12:7.2f} TFLOP/s (median)  {flops/ts[0]/1e12:7.2f} (best)")

if __name__ == "__main__":
    main()
