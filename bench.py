#!/usr/bin/env python3
"""bench.py — HelioField.render frames/s on MI355X (driver contract, see README).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg4] [--no-cpu]

A "step" is one ``HelioField.render`` forward over one batch of B synthetic sun positions
(B frames), inputs resident in HBM.  After the W warm-up steps a time-based preheat runs
(``--preheat`` seconds of the same call, default 0.5 — disclosed in the line as ``preheat_s``):
a fresh host path and a GPU coming out of idle run slow for their first ~0.1 s, and a K = 20
sample is only ≈0.1 ms long.  Then EXACTLY K steps are timed between barrier + synchronize
fences, MAX over ranks (with N > 1 the clock stops at each rank's own synchronize and the closing barrier
follows — the barrier's own latency is tens of µs beside a 91 µs sample; the reading with the clock stopped
behind it is in the line too, ``ms_per_step_with_closing_barrier``).  The process first binds all its threads — the runtime's helper threads too — to
one last-level-cache group (CCD) of its GPU's NUMA node (doodle_amd/affinity.py — what ``taskset`` would
do; measured: tools/core_sweep.py; disclosed as ``config.host_affinity``; the CPU baseline runs under the
original mask).

With N > 1 (``torch.distributed.run``, one rank per GPU) every rank renders its own B-row shard
of a global batch of N·B suns (weak scaling).  The suns are independent, so the timed loop has no
data-path collective (``value``); the same loop with the path's one collective — the RCCL
all-gather of image shards — is timed right after and always reported beside it
(``with_all_gather_every_step``; at config 2 it is interconnect-bound by construction, DESIGN.md
§5).  The multi-GPU figure of record of the path AS THE NORTH STAR STATES IT (shard B, all-gather
the images over xGMI) is ``multi_gpu_of_record``: every rank renders its 512-sun shard of
BASELINE config 5 (N=5000, R=256, 4096 suns on 8 GPUs) and all ranks gather all images every
step, the gather of step k overlapping the render of step k+1 on a side stream.  It runs
unchanged with one rank.  ``collective`` records the transport and the number of ranks RCCL
itself counts (ncclCommCount).  Rank 0 prints ONE JSON line.

Extra objects in the line (N = 1):
  roofline        the dominant kernel of the run by GPU time: the splat forward at BASELINE
                  config 4 (N=2000, B=512, R=512), timed live with HIP events on the launch
                  stream, ≥ 20 launches (f32 MFMA roofline; the HBM view is given beside it).
                  ``achieved`` / ``frac`` are those of the DENSE kernel (no device scratch: every (ray,
                  pixel) pair issued, a fraction of issued work ≤ 1); ``roofline.culled`` is the default
                  path — the rays that are exactly zero on a tile skipped, bit-identical images — with its
                  live fraction, time and speed-up; ``render_ms`` and the backward figures use the default.
                  ``traffic`` comes from the committed PMC passes (profiles/*_traffic.json) and is
                  tied to the kernel by name AND by the hash of its source file: a stale entry
                  is refused (null).
  roofline_bench_workload
                  the (launch-latency-bound) kernel of the timed config-2 loop, same method
  hbm_bound_kernels
                  the stages of the path that ARE HBM-bound (the loss block of HelioEnv.step and
                  the few-ray footprint backward), timed live at B=512, R=512: GB/s and fraction
                  of the 8 TB/s spec
  cpu_baseline    the oracle (oracle/torch_oracle.py, a CPU PyTorch restatement of the
                  reference that is bit-identical with it) timed on this host's cores, at the
                  best thread count of a short sweep
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from doodle_amd import HelioField  # noqa: E402
from doodle_amd import native, synthetic  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
F32_MFMA_PEAK_TF = 157.3     # dense f32 MFMA = f32 vector peak (spec)
TRAFFIC_FILES = ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")     # newest first
# the source file a kernel of the traffic files is built from: EVERY entry this file prints is checked against the hash
# of that file as it is now (measured_traffic), so a pass taken on other code is refused, not quoted
KERNEL_SOURCE = {"splat_fwd_mfma_tile": "splat_fwd.hip", "render_fwd_fused_small": "splat_fwd.hip",
                 "splat_fwd_mfma_tile(culled)": "splat_fwd.hip", "cull_fwd_kernel": "cull.hip",
                 "render_fwd_few": "splat_fwd.hip"}


def build_field(w, helios, errs, device, max_batch=None):
    f = HelioField(helios, synthetic.TARGET_POSITION, synthetic.TARGET_AREA, synthetic.TARGET_NORMAL,
                   error_scale_mrad=w.error_scale_mrad, sigma_scale=w.sigma_scale,
                   initial_action_noise=0.01, resolution=w.R, device=device,
                   max_batch_size=max_batch or errs.shape[0])
    f.batch_error_angles_mrad = errs.to(device)
    f.error_angles_mrad = errs[0].to(device)
    return f


def make_action(field, suns, noise):
    ideal = field.calculate_ideal_normals(suns)
    a = ideal + noise.to(ideal.device)
    a = a / a.norm(dim=2, keepdim=True)
    return a.reshape(a.shape[0], -1).contiguous()


def time_kernel(fn, iters, warm=3, repeats=1):
    """Average duration (s) of ``fn`` — one kernel launch — over ``iters`` back-to-back
    launches, bracketed by HIP events on the current (launch) stream.  ``repeats`` > 1 (the size
    sweeps under tools/, never a figure of the bench line): the least of that many such loops — a
    shared box now and then stalls one loop for tens of milliseconds (tools/sweep_render.py)."""
    for _ in range(warm):
        fn()
    best = None
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / iters
        best = t if best is None else min(best, t)
    return best


def time_interleaved(fns, iters, warm=2):
    """Average duration (s) of each of ``fns``, run in turn ``iters`` times — A, B, C, A, B, C … — every call
    bracketed by its own pair of HIP events on the launch stream: figures that are compared with each other
    (a kernel and the call that contains it) see the same clocks."""
    for _ in range(warm):
        for fn in fns:
            fn()
    pairs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)] for _ in fns]
    torch.cuda.synchronize()
    for i in range(iters):
        for k, fn in enumerate(fns):
            e0, e1 = pairs[k][i]
            e0.record()
            fn()
            e1.record()
    torch.cuda.synchronize()
    return [sum(e0.elapsed_time(e1) for e0, e1 in ps) * 1e-3 / iters for ps in pairs]


def preheat(fn, seconds, burst=None, fence=None, dist=None, device=None, clock=time.perf_counter):
    """Run ``fn`` for ``seconds`` of wall time.  With ``burst`` and ``fence``: in bursts of ``burst`` calls
    followed by ``fence()`` — the pattern of the timed region itself.  (Measured, tools/launch_jitter.py:
    after a long run of unsynchronised launches the first three or four "20 calls + synchronize" samples
    take 210–260 µs, then 156–168 µs for good; a preheat that never synchronises leaves the timed sample
    in that transient.)

    With a process group (``dist``) ``fn`` and ``fence`` contain collectives, so every rank must run the SAME
    number of bursts: whether to go on is agreed by all ranks (an all-reduce of each rank's own clock
    verdict) — a loop that every rank ends by its own clock runs one burst more on some ranks than on others
    whenever a burst ends within the ranks' start skew of the deadline, and the job hangs in the unmatched
    barrier (found by the two-rank rehearsal, ``--rehearse``: one run in six).  ``clock``: the wall clock (tests
    inject per-rank clocks that disagree about the deadline: tests/test_bench_preheat.py).  → bursts run."""
    t0 = clock()
    bursts = 0
    while True:
        go = clock() - t0 < seconds
        if dist is not None:
            verdict = torch.tensor([1 if go else 0], dtype=torch.int32, device=device)
            dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
            go = bool(verdict.item())
        if not go:
            return bursts
        bursts += 1
        if burst is None:
            fn()
        else:
            for _ in range(burst):
                fn()
            fence()


def source_sha16(name):
    with open(os.path.join(ROOT, "doodle_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def measured_traffic(kernel, N, B, R):
    """→ (HBM bytes per launch, note).  From the committed PMC passes, but only an entry taken
    for THIS kernel (by name) built from THIS source (hash of its .hip file, recorded with the
    pass by tools/pmc_traffic.py): bench.py cannot profile itself, and a number measured on other
    code is not evidence."""
    key = f"{kernel}@N={N},B={B},R={R}"
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                entry = json.load(f).get(key)
        except OSError:
            continue
        if entry is None:
            continue
        want = source_sha16(KERNEL_SOURCE[kernel])
        if entry.get("source_sha16") != want:
            return None, f"profiles/{name} holds {key} for source {entry.get('source_sha16')}, the library is built from {want}: stale, refused"
        return entry.get("hbm_bytes"), f"profiles/{name}, source {want}"
    return None, "no PMC pass committed for this kernel at this size"


def splat_roofline(field, suns, action, iters, variant=None, timer=None):
    """Roofline entry for the dominant kernel of one render on (field, suns): the fused
    render kernel when the problem takes the single-launch path, else the splat-forward
    kernel.  One launch per timed iteration, HIP events on the launch stream."""
    ops = native.get_ops()
    B, N, R = suns.shape[0], field.num_heliostats, field.resolution
    trig, stride = field._select_trig(B)
    normals = action.reshape(B, N, 3).contiguous()
    actual, _, rays = ops.geometry_fwd(field.heliostat_positions, suns, normals, trig, stride, field._plane)
    image = torch.empty((B, R, R), dtype=torch.float32, device=suns.device)
    lib, st = ops.lib, native._stream()
    var = ops.splat_variant if variant is None else variant
    fused = var in (0, 2) and lib.helio_render_fwd_launches(B, N, R) == 1
    if fused:
        kernel = "render_fwd_fused_small"
        args = (B, N, R, field.heliostat_positions.data_ptr(), suns.data_ptr(), normals.data_ptr(), trig.data_ptr(),
                stride, field._plane, field._xs.data_ptr(), field._ys.data_ptr(), actual.data_ptr(), None,
                rays.data_ptr(), image.data_ptr(), var, None, 0, st)
        t = time_kernel(lambda: lib.helio_render_fwd(*args), iters, warm=max(3, iters // 10))
        bytes_alg = 4.0 * B * R * R + 32.0 * B * N + 16.0 * B * N + 12.0 * N + 12.0 * B + 8.0 * R
    else:
        kernel = "splat_fwd"
        # no device scratch: the DENSE kernel, every (ray, pixel) pair issued — the roofline of record
        args = (B, N, R, rays.data_ptr(), field._xs.data_ptr(), field._ys.data_ptr(), image.data_ptr(), var, None, 0, st)
        t = time_kernel(lambda: lib.helio_splat_fwd(*args), iters) if timer is None else timer(lambda: lib.helio_splat_fwd(*args))
        bytes_alg = 4.0 * B * R * R + 16.0 * B * N + 8.0 * R   # image store + ray parameters + xs/ys
    flops = 2.0 * B * N * R * R                       # one FMA per (ray, pixel)
    traffic, note = measured_traffic(kernel if fused else "splat_fwd_mfma_tile", N, B, R)
    return {
        "bound": "mfma", "kernel": kernel, "achieved": round(flops / t / 1e12, 3),
        "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(flops / t / 1e12 / F32_MFMA_PEAK_TF, 4),
        "traffic": traffic, "traffic_source": note,
        "kernel_us": round(t * 1e6, 2), "kernel_launches_timed": iters,
        "hbm_achieved_GBs": round(bytes_alg / t / 1e9, 1), "hbm_frac": round(bytes_alg / t / 1e9 / HBM_PEAK_GBS, 4),
        "algorithmic_bytes": bytes_alg, "algorithmic_flops": flops,
    }


def sigma_pixels(field, suns, action):
    """Footprint sigma of the field's rays in pixels (min / median / max over the rays that hit the plane):
    sigma = sqrt(log2(e) / (2 k2)) from the ray parameters the geometry kernel writes, pixel pitch W / (R - 1)."""
    ops = native.get_ops()
    B, N = suns.shape[0], field.num_heliostats
    trig, stride = field._select_trig(B)
    _, _, rays = ops.geometry_fwd(field.heliostat_positions, suns, action.reshape(B, N, 3).contiguous(), trig, stride,
                                  field._plane)
    k2 = rays[..., 2].flatten()
    k2 = k2[k2 > 0]
    pitch = field.target_width / max(field.resolution - 1, 1)
    sig = torch.sqrt(1.4426950408889634 / (2.0 * k2)) / pitch
    return {"min": round(float(sig.min()), 3), "median": round(float(sig.median()), 3), "max": round(float(sig.max()), 3),
            "pixel_pitch_m": round(pitch, 5)}


def culled_splat(field, suns, action, variant):
    """The default splat call — device scratch handed over, exactly-zero rays skipped (csrc/cull.h) — as a closure
    for the timers, the fraction of (ray, tile) pairs it keeps, and whether its image equals the dense one's bits."""
    ops = native.get_ops()
    lib, st = ops.lib, native._stream()
    B, N, R = suns.shape[0], field.num_heliostats, field.resolution
    trig, stride = field._select_trig(B)
    _, _, rays = ops.geometry_fwd(field.heliostat_positions, suns, action.reshape(B, N, 3).contiguous(), trig, stride,
                                  field._plane)
    nb = lib.helio_fwd_scratch_bytes(B, N, R, variant)
    if nb <= 0:
        return None
    scratch = torch.empty(nb, dtype=torch.uint8, device=suns.device)
    image = torch.empty((B, R, R), dtype=torch.float32, device=suns.device)
    args = (B, N, R, rays.data_ptr(), field._xs.data_ptr(), field._ys.data_ptr(), image.data_ptr(), variant,
            scratch.data_ptr(), nb, st)
    native._check(lib, lib.helio_splat_fwd(*args))
    t = -(-R // 256)
    live = scratch[:4 * B * t * t].view(torch.int32).double().sum().item() / (B * t * t * N)
    dense = ops.splat_fwd(rays, field._xs, field._ys, variant=variant, cull=False)
    same = bool(torch.equal(image.view(torch.int32), dense.view(torch.int32)))
    return (lambda: lib.helio_splat_fwd(*args)), live, same, (scratch, image, rays)


def value_sweep(dev, w, steps, seeds=(0, 1, 2), sigmas=(0.01, 0.1)):
    """the headline loop — K bare ``field.render`` calls + fence, after a burst preheat — for seeds
    0..2 at the training sigma_scale (0.01) and at the README default (0.1): frames/s per seed, median, spread."""
    import gc
    # This sweep runs late in the process; round 3's line showed the SAME workload at 4.17 M here against 5.48 M as
    # `value`.  tools/headline_drift.py replays the headline sample after each thing the process does in between, one at
    # a time, on two boxes (profiles/r04_d_headline_drift.txt, r04_f_headline_drift.txt): the loop has TWO MODES — 4.35 µs
    # per step (5.7 M frames/s; device period of a long loop 3.72 µs) and 5.3–6.5 µs (3.8–4.7 M; device period 4.1–4.2 µs,
    # i.e. the dispatch itself is slower, not the host's issue) — and drops into the slow one at DIFFERENT points in the
    # two runs (once when a side stream had run, and again after 20 GB had been allocated and freed; once after
    # config-4-sized launches, then for every later sample) and comes back by itself or — every time — when the threads'
    # affinity is reset (widened to the NUMA node and bound to the CCD again).  None of the listed state is the cause
    # (allocator pool, streams, graphs, a second field, HelioEnv, the autograd thread: each neutral in one of the runs);
    # it is where threads and queue lines sit after a phase that moved them (tools/core_sweep.py: the CCD the runtime's
    # helper threads live on is 25 % faster than its neighbours).  So: re-bind before sampling, and say so.
    gc.collect()
    reserved0 = torch.cuda.memory_reserved(dev)
    torch.cuda.empty_cache()
    from doodle_amd import affinity
    affinity.widen_to_node(dev.index or 0)
    rebound = affinity.bind_to_gpu_ccd(dev.index or 0)
    out = {"steps": steps, "what": "frames/s of K field.render calls + synchronize (the timed region of `value`), one fresh "
                                   "field per (seed, sigma_scale); spread = (max - min) / median",
           "before_sampling": {"threads_rebound_to": rebound["l3_group"] if rebound else None,
                               "allocator_GB_reserved_then_released": round(reserved0 / 2**30, 2),
                               "why": "the launch-bound loop has two modes (4.35 / 5.3–6.5 µs per step) and a phase that moves "
                                      "threads can leave it in the slow one until the affinity is reset — "
                                      "profiles/r04_d_headline_drift.txt, r04_f_headline_drift.txt; round 3 sampled without "
                                      "resetting it (seed 0 here = the workload of `value`: 4.17 M then against 5.48 M)"}}
    for sg in sigmas:
        vals = []
        for seed in seeds:
            ws = synthetic.Workload(w.name, w.N, w.B, w.R, sg, w.error_scale_mrad, w.span)
            helios, suns, errs, noise = synthetic.make_inputs(ws, seed)
            f = build_field(ws, helios, errs, dev)
            suns_d = suns.to(dev)
            a = make_action(f, suns_d, noise)
            with torch.no_grad():
                preheat(lambda: f.render(suns_d, a, None), 0.25, burst=steps, fence=torch.cuda.synchronize)
                gc_was = gc.isenabled()
                gc.disable()
                try:
                    best = float("inf")
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(steps):
                            f.render(suns_d, a, None)
                        torch.cuda.synchronize()
                        best = min(best, time.perf_counter() - t0)
                finally:
                    if gc_was:
                        gc.enable()
            vals.append(round(ws.B * steps / best, 1))
        med = sorted(vals)[len(vals) // 2]
        out[f"sigma_scale={sg}"] = {"frames_per_s_by_seed": dict(zip(map(str, seeds), vals)), "median": med,
                                    "spread": round((max(vals) - min(vals)) / med, 4)}
    return out


def cpu_baseline_extra(seed):
    """the oracle on this host at config 1 (N=50, B=1: the reference's own CPU-runnable case), forward,
    and at config 3 (N=50, B=25) forward + backward (L = (img*G).sum() + actual.sum(), torch autograd)."""
    from oracle import torch_oracle as to
    out = {}
    threads0 = torch.get_num_threads()
    try:
        torch.set_num_threads(min(8, os.cpu_count() or 8))
        for key, cfg, bwd in (("cfg1_forward", "cfg1", False), ("cfg3_forward_backward", "cfg2", True)):
            w = synthetic.CONFIGS[cfg]
            helios, suns, errs, noise = synthetic.make_inputs(w, seed)
            sc = to.Scene.build(helios, synthetic.TARGET_POSITION, synthetic.TARGET_AREA, synthetic.TARGET_NORMAL, w.R, w.sigma_scale)
            ideal = to.ideal_normals(helios, sc.target_position, suns)
            a = ideal + noise
            a = (a / a.norm(dim=2, keepdim=True)).reshape(w.B, -1)
            e = errs if w.B > 1 else errs[:1]
            G = torch.randn(w.B, w.R, w.R, generator=torch.Generator().manual_seed(0))

            def call():
                if not bwd:
                    with torch.no_grad():
                        to.render(sc, suns, a, e)
                    return
                x = a.clone().requires_grad_(True)
                img, actual = to.render(sc, suns, x, e)
                torch.autograd.grad((img * G).sum() + actual.sum(), x)

            call()
            n, t0 = 0, time.perf_counter()
            while True:
                call()
                n += 1
                el = time.perf_counter() - t0
                if el > 3.0 or n >= 40:
                    break
            out[key] = {"frames_per_s": round(w.B * n / el, 2), "ms_per_call": round(el / n * 1e3, 2), "calls": n,
                        "threads": torch.get_num_threads(), "workload": w.name + (" forward+backward" if bwd else " forward")}
    finally:
        torch.set_num_threads(threads0)
    return out


def cpu_baseline(w, seed, budget_s=10.0):
    """The oracle timed on this host: forward render of the bench workload, at the best thread
    count of a short sweep (a 256-thread host is slower with all its threads than with a few:
    the reference's ATen kernels are bound by the [M,R,R,3] temporaries, not by arithmetic)."""
    from oracle import torch_oracle as to
    helios, suns, errs, noise = synthetic.make_inputs(w, seed)
    sc = to.Scene.build(helios, synthetic.TARGET_POSITION, synthetic.TARGET_AREA, synthetic.TARGET_NORMAL,
                        w.R, w.sigma_scale)
    ideal = to.ideal_normals(helios, sc.target_position, suns)
    a = ideal + noise
    a = (a / a.norm(dim=2, keepdim=True)).reshape(w.B, -1)
    threads0 = torch.get_num_threads()
    ncpu = os.cpu_count() or threads0
    sweep = {}
    try:
        with torch.no_grad():
            for t in sorted({t for t in (8, 16, 32, 64, 128, threads0) if t <= ncpu}):
                torch.set_num_threads(t)
                to.render(sc, suns, a, errs)                     # warm-up at this thread count
                t0 = time.perf_counter()
                for _ in range(2):
                    to.render(sc, suns, a, errs)
                sweep[t] = round(2 * w.B / (time.perf_counter() - t0), 2)
            best = max(sweep, key=sweep.get)
            torch.set_num_threads(best)
            to.render(sc, suns, a, errs)
            n, t0 = 0, time.perf_counter()
            while True:
                to.render(sc, suns, a, errs)
                n += 1
                el = time.perf_counter() - t0
                if el > budget_s or n >= 60:
                    break
    finally:
        torch.set_num_threads(threads0)
    return {"value": round(w.B * n / el, 2), "unit": "frames/s", "cores": best,
            "kind": "port", "host_cpus": ncpu, "thread_sweep_frames_per_s": {str(k): v for k, v in sweep.items()},
            "sample": f"{n} forward renders of {w.name} (all {w.B} suns, whole workload) at {best} threads, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 20000 at the microsecond-scale configs, 50 at cfg4, 10 at cfg5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps first (default: a tenth of --steps)")
    ap.add_argument("--preheat", type=float, default=0.5,
                    help="seconds of untimed calls between the warm-up steps and the timed steps (disclosed as preheat_s)")
    ap.add_argument("--workload", default="cfg2", choices=sorted(synthetic.CONFIGS))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-large", action="store_true", help="skip the config-4 roofline leg")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary legs (fwd+bwd, env.step, HBM-bound kernels, config-5 shard): used for "
                         "profiling runs so that every kernel in the trace belongs to one workload")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "fwdbwd"])
    ap.add_argument("--overlap", action="store_true",
                    help="config-2 loops: run the all-gather on a side stream (pays for steps of milliseconds; at "
                         "config 2 the cross-stream events cost more than they hide)")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="all-gather the images inside the timed loop (default: images stay on their rank; the "
                         "gathered loop is timed afterwards and reported as with_all_gather_every_step)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and all-gather even with one rank (testing)")
    ap.add_argument("--rehearse", action="store_true",
                    help="(development, no multi-GPU box at hand) walk the N > 1 control flow on ONE GPU: every rank uses "
                         "cuda:0, the process group is gloo and the gather goes through torch.distributed.  The line "
                         "carries \"rehearsal\": true — its figures mean nothing")
    ap.add_argument("--cfg5-suns", type=int, default=512, help="suns per GPU of the config-5 leg (BASELINE: 4096 / 8)")
    ap.add_argument("--cfg5-steps", type=int, default=20)
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"cfg4": 50, "cfg5": 10}.get(args.workload, 20000)
    if args.warmup is None:
        args.warmup = max(2, args.steps // 10)

    # stdout carries the ONE JSON line and nothing else: libraries that print to fd 1 (RCCL's
    # version banner at communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; before HIP initialises
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # launch-bound loop: keep the launching thread on the socket the card hangs off (doodle_amd/affinity.py);
    # the CPU baseline below runs under the original mask
    from doodle_amd import affinity
    cpu_mask0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    torch.empty(1, device=dev)                     # the runtime's helper threads exist from here on
    numa = affinity.bind_to_gpu_ccd(local)         # slot = the card's index among the cards of its NUMA node
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)          # only matters for a bare --force-dist run
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    w = synthetic.CONFIGS[args.workload]
    helios, suns, errs, noise = synthetic.make_inputs(w, args.seed, b_offset=rank * w.B, b_count=w.B)
    field = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    action = make_action(field, suns_d, noise)
    gather, gathered, gather_error = None, None, None

    def make_gather():
        """The image all-gather (RCCL through libhelio_comm.so) and its two output buffers; all ranks get one or none
        does.  If the communicator cannot be created the gathered legs are skipped and said so."""
        nonlocal gather, gathered, gather_error
        try:
            from doodle_amd.comm import ImageGather
            gather = ImageGather(transport="torch" if args.rehearse else "auto")
            gathered = [torch.empty((world * w.B, w.R, w.R), dtype=torch.float32, device=dev) for _ in range(2)]
        except Exception as e:  # noqa: BLE001
            gather, gather_error = None, repr(e)
        flag = torch.tensor([0 if gather is not None else 1], device=dev)
        dist.all_reduce(flag)
        if int(flag.item()) != 0 and gather is not None:
            gather.close()
            gather, gather_error = None, "another rank could not create the communicator"

    # The headline loop has no collective, so the communicator is made AFTER it unless the loop itself gathers
    # (--gather-every-step): a second RCCL instance in the process (its proxy threads beside the launching thread, on
    # the one CCD all threads are bound to) costs a launch-bound loop 20–45 % (one box, K = 20, five runs each: 5.0–5.7 M
    # frames/s without it, 3.1–4.4 M with it, 5.1–5.7 M with only torch.distributed's own communicator).
    if dist is not None and args.gather_every_step:
        make_gather()
    stepno = [0]
    if args.mode == "fwdbwd":
        action.requires_grad_(True)
        G = torch.ones((w.B, w.R, w.R), device=dev)

    def step():
        if args.mode == "fwd":
            with torch.no_grad():
                img, _ = field.render(suns_d, action, None)
        else:
            img, actual = field.render(suns_d, action, None)
            torch.autograd.grad((img * G).sum() + actual.sum(), action)
        if gather is not None and gather_now[0]:
            # stream-ordered behind the render by default; --overlap puts it on a side stream
            gather.gather(img.detach(), gathered[stepno[0] & 1], overlap=args.overlap)
            stepno[0] += 1

    def fence():
        if gather is not None:
            gather.wait()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def stop_clock(t0):
        """End of a timed region: the clock stops when THIS rank's K steps are done (its collectives waited for, its
        device synchronised); the closing barrier + synchronize of the bracket follow, and the job's time is the MAX
        over ranks — the moment the slowest rank was done.  With the clock stopped behind the barrier instead, a 91 µs
        sample would mostly measure the barrier's own latency (tens of µs over 8 ranks, none at all with one rank);
        that reading is reported beside it (``ms_per_step_with_closing_barrier``).  → (seconds, seconds incl. barrier)"""
        if gather is not None:
            gather.wait()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is None:
            return el, el
        dist.barrier()
        torch.cuda.synchronize()
        el_b = time.perf_counter() - t0
        t = torch.tensor([el, el_b], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()), float(t[1].item())

    gather_now = [bool(args.gather_every_step)]
    for _ in range(args.warmup):
        step()
    # disclosed (preheat_s): host path and GPU clocks at steady state, in bursts of K steps + fence like
    # the timed region (at least four of them)
    preheat(step, args.preheat, burst=args.steps, fence=fence, dist=dist, device=dev)
    for _ in range(4 if args.steps <= 1000 else 0):
        for _ in range(args.steps):
            step()
        fence()
    fence()
    if args.mode == "fwd" and not gather_now[0]:
        # the timed loop proper: K calls of HelioField.render and nothing else (config 2 is ≈5 µs of
        # GPU per call, so a closure call and a no_grad() enter/exit per step would be ≈15 % of it)
        import gc
        render, K = field.render, args.steps
        gc_was = gc.isenabled()
        gc.disable()                     # as timeit does: tensors are reference-counted, the cycle collector only adds jitter
        try:
            with torch.no_grad():
                t0 = time.perf_counter()
                for _ in range(K):
                    render(suns_d, action, None)
                el, el_barrier = stop_clock(t0)
        finally:
            if gc_was:
                gc.enable()
    else:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        el, el_barrier = stop_clock(t0)

    # the same shards with the other treatment of the images (gathered every step / left on their
    # rank): at config 2 the gathered loop is bound by delivering (N-1) x 1.64 MB to every rank per
    # step, DESIGN.md §5 — always reported beside `value`, never instead of it
    el_other = None
    if dist is not None and gather is None and gather_error is None:
        make_gather()
    if gather is not None and numa is not None:
        # the gathered legs are not launch-bound and run RCCL's proxy threads: the whole NUMA node for them
        affinity.widen_to_node(local)
    if gather is not None:
        gather_now[0] = not gather_now[0]
        for _ in range(min(args.warmup, 50)):
            step()
        preheat(step, min(args.preheat, 0.2), burst=args.steps, fence=fence, dist=dist, device=dev)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        el_other, _ = stop_clock(t0)
        gather_now[0] = not gather_now[0]

    # the multi-GPU figure of record, on EVERY rank (collective): the real per-GPU shard of BASELINE
    # config 5, rendered AND all-gathered every step — see DESIGN.md §5
    shard = None
    if not args.no_large and not args.no_extras:
        try:
            shard = cfg5_shard_leg(dev, rank, world, gather, dist, args.seed, args.cfg5_suns, args.cfg5_steps)
        except Exception as e:  # noqa: BLE001
            shard = {"error": repr(e)}

    if rank == 0:
        frames = world * w.B * args.steps
        out = {
            "metric": "HelioField.render frames/sec", "value": round(frames / el, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preheat_s": args.preheat,
            "ms_per_step": round(el / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            **({"ms_per_step_with_closing_barrier": round(el_barrier / args.steps * 1e3, 5)} if world > 1 else {}),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({"rehearsal": True, "rehearsal_note": "all ranks on ONE GPU over gloo: control flow only, no figure of this line is a measurement"}
               if args.rehearse else {}),
            "config": {"workload": f"{w.name} {'forward' if args.mode == 'fwd' else 'forward+backward'} "
                                   f"HelioField.render, sigma_scale={w.sigma_scale}, err={w.error_scale_mrad} mrad, "
                                   f"{w.B} suns per GPU",
                       "global_batch": world * w.B, "parallelism": f"sun-batch sharded x{world}"
                       + ((f", RCCL all-gather of images every step ({gather.transport} transport, "
                           f"{'side stream' if args.overlap else 'stream-ordered'})") if (gather is not None and args.gather_every_step)
                          else (", no data-path collective (images stay on the rank that rendered them)" if world > 1 else "")),
                       "timing": f"{args.warmup} warm-up steps, {args.preheat} s time-based preheat (untimed, in bursts of "
                                 f"{args.steps} steps + fence), then {args.steps} timed steps between barrier + synchronize "
                                 "fences, max over ranks" + ("; the clock stops at each rank's own synchronize, the closing barrier "
                                 "follows (ms_per_step_with_closing_barrier: the clock stopped behind it)" if world > 1 else ""),
                       "host_affinity": (f"all threads of the process bound to one last-level-cache group (CPUs {numa['l3_group']}, "
                                         f"{numa['cpus']} of them) of the GPU's NUMA node {numa['numa_node']}"
                                         if numa is not None else "scheduler's choice")},
        }
        if dist is not None:
            out["collective"] = {
                "transport": gather.transport if gather is not None else None,
                "rccl_ranks": gather.rccl_ranks if gather is not None else None,       # ncclCommCount of the communicator
                "torch_distributed_backend": dist.get_backend(), "world_size": dist.get_world_size()}
        if el_other is not None:
            key = "without_all_gather" if args.gather_every_step else "with_all_gather_every_step"
            out[key] = {"frames_per_s": round(frames / el_other, 1), "ms_per_step": round(el_other / args.steps * 1e3, 5),
                        "MB_received_per_rank_and_step": round((world - 1) * w.B * w.R * w.R * 4 / 1e6, 2),
                        "note": ("same shards, images left on the rank that rendered them" if args.gather_every_step else
                                 f"same shards, every image delivered to every rank each step ({gather.transport} transport): "
                                 "interconnect-bound at this frame size, DESIGN.md §5")}
        if gather_error is not None:
            out["all_gather_error"] = gather_error
        if shard is not None:
            out["multi_gpu_of_record"] = shard
            if "frames_per_s" in shard:
                # lifted next to `value`: `value` (BASELINE's metric, config 2) has NO data-path collective and scales
                # N x by construction; the figure that exercises the interconnect is this one
                out["scaling_of_record"] = {"frames_per_s": shard["frames_per_s"], "ms_per_step": shard["ms_per_step"],
                                            "n_gpus": world, "rccl_ranks": shard.get("rccl_ranks"),
                                            "workload": shard["workload"]}
                out["config"]["scaling_curve"] = ("`value` is BASELINE's metric (config 2, images stay on their rank: no "
                                                  "collective, N x by construction); a scaling curve of the path AS THE NORTH "
                                                  "STAR STATES IT (shard B, all-gather the images over xGMI) must use "
                                                  "scaling_of_record.frames_per_s (config 5 shard, render + all-gather every step)")
        iters = 2000 if w.B * w.N * w.R * w.R < 1e10 else 20
        small = splat_roofline(field, suns_d, action.detach(), iters)
        small["workload"] = w.name
        out["roofline"] = small
        if world == 1:
            if not args.no_large and args.workload != "cfg4":
                # The dominant kernel of this run by GPU time (profiles/: >90 %) is the splat
                # forward at the large-field configuration — that is the roofline entry; the
                # launch-latency-bound kernel of the timed config-2 loop is reported beside it.
                try:
                    out["roofline"] = large_leg(dev, args.seed)
                    out["roofline_bench_workload"] = small
                except Exception as e:  # noqa: BLE001  (report, do not hide)
                    out["roofline_large_error"] = repr(e)
            if not args.no_large and not args.no_extras:
                try:
                    out["extras"] = extras_leg(field, suns_d, action.detach(), w, dev)
                except Exception as e:  # noqa: BLE001
                    out["extras"] = {"error": repr(e)}
                try:
                    out["hbm_bound_kernels"] = hbm_leg(dev)
                except Exception as e:  # noqa: BLE001
                    out["hbm_bound_kernels"] = {"error": repr(e)}
            if not args.no_large and not args.no_extras and args.mode == "fwd":
                try:
                    out["value_by_seed_and_sigma"] = value_sweep(dev, w, args.steps if args.steps <= 1000 else 1000)
                    out["sigma_px"] = {w.name: sigma_pixels(field, suns_d, action.detach())}
                except Exception as e:  # noqa: BLE001
                    out["value_by_seed_and_sigma"] = {"error": repr(e)}
            if not args.no_cpu:
                if numa is not None:
                    affinity.restore(cpu_mask0)
                out["cpu_baseline"] = cpu_baseline(w, args.seed)
                out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
                if not args.no_extras:
                    try:
                        out["cpu_baseline_other_configs"] = cpu_baseline_extra(args.seed)
                    except Exception as e:  # noqa: BLE001
                        out["cpu_baseline_other_configs"] = {"error": repr(e)}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()                 # rank 0 runs extra single-GPU legs; tear down together
    if gather is not None:
        gather.close()
    if dist is not None:
        dist.destroy_process_group()


def extras_leg(field, suns_d, action, w, dev):
    """Secondary figures SURVEY.md §8(d) asks for, at the bench workload: render forward+backward
    (config 3) and HelioEnv.step forward, both through the Python surface (wall clock)."""
    from doodle_amd.env import HelioEnv

    def wall(fn, n, repeats=3):
        preheat(fn, 0.2)                            # a fresh Python path runs slow for its first ~0.1 s
        best = float("inf")
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        return best

    a = action.clone().requires_grad_(True)
    G = torch.randn((w.B, w.R, w.R), device=dev)

    def fwdbwd():
        img, actual = field.render(suns_d, a, None)
        torch.autograd.grad((img * G).sum() + actual.sum(), a)

    # BASELINE config 3 — forward + backward through the render for L = (img·G).sum() + actual.sum() —
    # three ways, same kernels, same numbers (GPU tests):
    #   render_fwd_bwd_us           HelioField.render_value_and_grad: forward and backward kernels enqueued
    #                               back to back by ONE binding call for the cotangents (G, 1) — no autograd graph
    #   render_fwd_bwd_autograd_us  render + the loss as torch ops + torch.autograd.grad (the autograd engine's
    #                               thread hand-off and the bench's own torch ops are most of it)
    #   render_fwd_bwd_graph_us     the autograd iteration captured once and replayed as a HIP graph
    ones = torch.ones((w.B, w.N, 3), device=dev)
    t_vg = wall(lambda: field.render_value_and_grad(suns_d, action, G, ones), 300)
    preheat(fwdbwd, 0.7)        # the first differentiating loop of a process runs 1.5-2x slow for ~0.5 s (tools/bench_env.py)
    t_fb = wall(fwdbwd, 300)
    out = {"render_fwd_bwd_frames_per_s": round(w.B / t_vg, 1), "render_fwd_bwd_us": round(t_vg * 1e6, 1),
           "render_fwd_bwd_autograd_us": round(t_fb * 1e6, 1)}
    try:
        from doodle_amd.graphed import GraphedRenderGrad
        gr = GraphedRenderGrad(field, suns_d, like=action, loss=lambda img, actual: (img * G).sum() + actual.sum())
        out["render_fwd_bwd_graph_us"] = round(wall(lambda: gr(), 300) * 1e6, 1)
    except Exception as e:  # noqa: BLE001
        out["render_fwd_bwd_graph_error"] = repr(e)
    # the driver's own sample — 20 renders and a fence — as ONE HIP graph (captured once, untimed): what a caller who captures
    # its loop gets on any host; the eager sample (`value` at --steps 20) is bound by the host's launch rate and the fence's
    # wake-up, 4.3–5.7 M frames/s by box
    try:
        side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(3):
                field.render(suns_d, action, None)
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad(), torch.cuda.graph(graph):
            kept = [field.render(suns_d, action, None)[0] for _ in range(20)]
        t_g = wall(graph.replay, 50)
        out["render_graph_of_20_frames_per_s"] = round(20 * w.B / t_g, 1)
        out["render_graph_of_20_us_per_step"] = round(t_g / 20 * 1e6, 2)
        del kept, graph
    except Exception as e:  # noqa: BLE001
        out["render_graph_of_20_error"] = repr(e)
    env = HelioEnv(field.heliostat_positions, torch.tensor(synthetic.TARGET_POSITION, device=dev), synthetic.TARGET_AREA,
                   torch.tensor(synthetic.TARGET_NORMAL, device=dev), sigma_scale=w.sigma_scale,
                   error_scale_mrad=w.error_scale_mrad, resolution=w.R, batch_size=w.B, device=dev)
    env.set_sun_pos(suns_d)
    env.reset()
    with torch.no_grad():
        t_step = wall(lambda: env.step(action), 300)

    def step_bwd():          # the reference's test-time-compute iteration: step + dist.backward()
        _, m, _ = env.step(a)
        m["dist"].backward()
        a.grad = None

    t_sb = wall(step_bwd, 300)
    out.update({
        "env_step_fwd_frames_per_s": round(w.B / t_step, 1), "env_step_fwd_us": round(t_step * 1e6, 1),
        "env_step_fwd_bwd_us": round(t_sb * 1e6, 1),
        "note": "config 3 (L = (img*G).sum()+actual.sum()): render_fwd_bwd_us = HelioField.render_value_and_grad, "
                "forward + backward kernels for the cotangents (G, 1) in one binding call; _autograd_us = render + "
                "torch ops + torch.autograd.grad; _graph_us = that iteration replayed from a HIP graph.  env.step = "
                "render + loss block (2 launches) + NaN/Inf check (one wait on a pinned host record); "
                "env_step_fwd_bwd adds metrics['dist'].backward(); best of 3 wall-clock loops through the Python surface.  "
                "render_graph_of_20: 20 field.render calls captured once as a HIP graph, one replay + synchronize per sample"})
    return out


def hbm_leg(dev, B=512, N=2000, R=512, iters=20):
    """The stages of the path that ARE HBM-bound, timed live (HIP events, ``iters`` launches each):
    the loss block of HelioEnv.step (test_environment.py:436-457: 12 B per pixel read once) forward
    and backward, and the few-ray footprint backward (the reference's TTT sweeps run N = 1) alone and
    with the loss adjoint formed on the fly.  GB/s of algorithmic bytes against the 8 TB/s spec."""
    import ctypes
    from doodle_amd.losses import StepConstants
    g = torch.Generator(device=dev).manual_seed(0)
    img, target, dm = (torch.rand(B, R, R, device=dev, generator=g) for _ in range(3))
    unit = lambda t: t / t.norm(dim=-1, keepdim=True)  # noqa: E731
    ideal = unit(torch.rand(B, N, 3, device=dev, generator=g))
    actual = unit(ideal + 0.01 * torch.rand(B, N, 3, device=dev, generator=g))
    action = unit(ideal + 0.1 * torch.rand(B, N, 3, device=dev, generator=g))
    helios = torch.rand(N, 3, device=dev, generator=g) * 10 + 80
    f3 = ctypes.c_float * 3
    c = StepConstants(target, target.amax((1, 2)).clamp_min(1e-6), dm, ideal, helios, f3(0, -5, 0), f3(0, 1, 0), 15.0, 15.0, False)
    ops = native.get_ops()
    one = torch.ones((), device=dev)

    def entry(t, nbytes, what):
        return {"us": round(t * 1e6, 1), "GBs": round(nbytes / t / 1e9, 1), "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
                "algorithmic_bytes": nbytes, "bytes": what}

    out = {"workload": f"B={B}, N={N}, R={R}", "peak_GBs": HBM_PEAK_GBS, "launches_timed": iters}
    t = time_kernel(lambda: ops.step_losses_fwd(img, actual, action, c), iters)
    out["step_losses_fwd"] = entry(t, 12.0 * B * R * R + 48.0 * B * N, "img + target + distance map read once, 48 B per ray")
    t = time_kernel(lambda: ops.step_losses_bwd(img, actual, action, c, one, one, one, one, None, True, True, True), iters)
    out["step_losses_bwd"] = entry(t, 16.0 * B * R * R + 60.0 * B * N, "the same + the image cotangent written")
    Nf = 1
    rays = (torch.rand(B, Nf, 4, device=dev, generator=g) * torch.tensor([10., 10., 0.5, 0.01], device=dev)
            - torch.tensor([5., 5., 0., 0.], device=dev))
    xs = torch.linspace(-7.5, 7.5, R, device=dev)
    ys = xs.clone()
    Gi = torch.randn(B, R, R, device=dev, generator=g)
    t = time_kernel(lambda: ops.splat_bwd(rays, xs, ys, Gi, variant=4), iters)
    out["splat_bwd_few(N=1)"] = entry(t, 4.0 * B * R * R, "the image cotangent read once")
    lib = ops.lib
    mom = torch.empty(B, lib.helio_splat_bwd_blocks(R), Nf, 5, device=dev)
    grad = torch.empty(B, Nf, 3, device=dev)
    plane = native.Plane()
    plane.origin[:], plane.normal[:], plane.u[:], plane.v[:], plane.w[:] = (0, -5, 0), (0, 1, 0), (1, 0, 0), (0, 0, 1), (0, -1, 0)
    plane.sigma_scale = 0.01
    sun = torch.rand(B, 3, device=dev, generator=g) * 1e4
    act1 = unit(torch.rand(B, Nf, 3, device=dev, generator=g))
    trig = torch.tensor([1., 0., 1., 0.], device=dev).repeat(B, Nf, 1).contiguous()
    ideal1 = unit(torch.rand(B, Nf, 3, device=dev, generator=g))
    h1 = helios[:Nf].contiguous()

    def fused():
        native._check(lib, lib.helio_env_step_bwd(
            B, Nf, R, h1.data_ptr(), sun.data_ptr(), act1.data_ptr(), trig.data_ptr(), 4 * Nf, plane, rays.data_ptr(),
            xs.data_ptr(), ys.data_ptr(), img.data_ptr(), target.data_ptr(), c.tx.data_ptr(), dm.data_ptr(),
            ideal1.data_ptr(), c.tp, c.tn, 15.0, 15.0, 0, None, one.data_ptr(), None, None, None, None, None, None,
            mom.data_ptr(), grad.data_ptr(), 0, None, 0, native._stream()))

    t = time_kernel(fused, iters)
    out["env_step_bwd_few_ray(N=1)"] = entry(t, 12.0 * B * R * R, "img + target + distance map read once (2 launches)")
    # … and the footprint FORWARD where it is HBM-bound: a handful of rays per image (the reference's test-time-compute
    # sweeps run ONE heliostat and 500 suns, run_experiments.py:31-56) — render_fwd_few, the whole render in one launch,
    # bound by writing the image once; north_star's ">= 60 % of peak HBM" has its measured counterpart here
    for Bf, Nf2, Rf in ((B, 1, R), (B, 8, R), (500, 1, 128)):
        wf = synthetic.Workload("few", N=Nf2, B=Bf, R=Rf, sigma_scale=0.02, error_scale_mrad=40.0)
        hf, sf, ef, nf = synthetic.make_inputs(wf, 0)
        ff = build_field(wf, hf, ef, dev)
        sfd = sf.to(dev)
        af = make_action(ff, sfd, nf).reshape(Bf, Nf2, 3).contiguous()
        tg, sd = ff._select_trig(Bf)
        with torch.no_grad():
            ws = ops.render_fwd(ff.heliostat_positions, sfd, af, tg, sd, ff._plane, ff._xs, ff._ys, want_refl=False)[3]
            xs_f, ys_f, pl_f = ff._xs, ff._ys, ff._plane
            t = time_kernel(lambda: ops.render_fwd(ff.heliostat_positions, sfd, af, tg, sd, pl_f, xs_f, ys_f, want_refl=False,
                                                   rays=ws, variant=13), max(iters, 50 if Bf * Rf * Rf < 1e8 else iters))
        key = f"render_fwd_few(B={Bf},N={Nf2},R={Rf})"
        out[key] = entry(t, 4.0 * Bf * Rf * Rf + 44.0 * Bf * Nf2, "the image written once, 44 B per ray (one launch = the whole render)")
        tr, note = measured_traffic("render_fwd_few", Nf2, Bf, Rf)
        if tr is not None or "stale" in note:
            out[key]["traffic"], out[key]["traffic_source"] = tr, note
    return out


def cfg5_shard_leg(dev, rank, world, gather, dist, seed, b_local=512, steps=20):
    """Every rank renders its own ``b_local``-sun shard of BASELINE config 5 (N=5000, R=256; 512 per
    GPU = 4096 over 8 GPUs) and ALL ranks gather ALL images every step; whole-job frames/s.  The
    gather of step k runs on a side stream and overlaps the render of step k+1 (two output buffers,
    two-deep back-pressure).  Compute per step ≈ 2.6 ms against 134 MB sent per rank."""
    w5 = synthetic.CONFIGS["cfg5"]
    w = synthetic.Workload(w5.name, w5.N, b_local, w5.R, w5.sigma_scale, w5.error_scale_mrad, w5.span)
    helios, suns, errs, noise = synthetic.make_inputs(w, seed, b_offset=rank * b_local, b_count=b_local)
    field = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    action = make_action(field, suns_d, noise)
    if gather is None and dist is None:
        # one process, no process group: the gather degenerates to what an all-gather over one rank
        # does, a device copy of the shard into the output buffer — the leg runs unchanged
        from doodle_amd.comm import ImageGather
        gather = ImageGather()
    out = None
    if gather is not None:
        out = [torch.empty((world * b_local, w.R, w.R), dtype=torch.float32, device=dev) for _ in range(2)]
    overlap = gather is not None and gather.transport == "rccl"

    def step():
        with torch.no_grad():
            img, _ = field.render(suns_d, action, None)
        if gather is not None:
            gather.gather(img, out[step.k & 1], overlap=overlap)
            step.k += 1

    step.k = 0

    def fence():
        if gather is not None:
            gather.wait()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(3):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0             # (the clock stops at the rank's own synchronize; the closing barrier follows)
    if dist is not None:
        dist.barrier()
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    try:
        sig = sigma_pixels(field, suns_d, action)
    except Exception as e:  # noqa: BLE001
        sig = {"error": repr(e)}
    return {"sigma_px": sig,
            "workload": f"BASELINE config 5 shard: N={w.N}, R={w.R}, {b_local} suns per GPU "
                        f"(global batch {world * b_local}), render + all-gather of all images every step",
            "n_gpus": world, "steps": steps, "warmup": 3,
            "frames_per_s": round(world * b_local * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 3),
            "collective": (f"{gather.transport} all-gather, {'side stream, overlapped with the next render' if overlap else 'stream-ordered'}"
                           if gather is not None else "none (the communicator could not be created)"),
            "rccl_ranks": gather.rccl_ranks if gather is not None else None,
            "MB_sent_per_rank_and_step": round(b_local * w.R * w.R * 4 / 1e6, 1),
            "MB_received_per_rank_and_step": round((world - 1) * b_local * w.R * w.R * 4 / 1e6, 1)}


def large_leg(dev, seed, iters=20):
    """Splat-forward roofline and whole-render rate at BASELINE config 4.  The dense kernel (the roofline of
    record), the default culled splat call and the whole ``field.render`` are timed in ONE interleaved,
    event-bracketed loop (A, B, C, A, B, C …), so the three figures see the same clocks; the backward
    (``render_value_and_grad``: forward + backward kernels for given cotangents) in a second one."""
    w = synthetic.CONFIGS["cfg4"]
    helios, suns, errs, noise = synthetic.make_inputs(w, seed)
    field = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    action = make_action(field, suns_d, noise)
    ops = native.get_ops()
    timed = {}

    def render():
        field.render(suns_d, action, None)

    culled = culled_splat(field, suns_d, action, 5)

    def timer(dense_call):
        with torch.no_grad():
            ts = time_interleaved([dense_call, render] + ([culled[0]] if culled else []), iters)
        timed["render"], timed["culled"] = ts[1], (ts[2] if culled else None)
        return ts[0]

    r = splat_roofline(field, suns_d, action, iters=iters, variant=5, timer=timer)
    el = timed["render"]
    if culled:
        tc, live = timed["culled"], culled[1]
        r["culled"] = {
            "what": "the default path: helio_splat_fwd with device scratch — per 256x256 tile, the rays whose every product "
                    "underflows below half an ulp of any accumulator are compacted out first (cull_fwd_kernel + "
                    "cull_order_fwd_kernel + splat_fwd_mfma_tile<4> on the lists); csrc/cull.h",
            "live_fraction": round(live, 4), "splat_us": round(tc * 1e6, 2),
            "speedup_vs_dense_kernel": round(r["kernel_us"] / (tc * 1e6), 3),
            "frac_on_live_flops": round(live * r["algorithmic_flops"] / tc / 1e12 / F32_MFMA_PEAK_TF, 4),
            "dense_equivalent_TFLOPs": round(r["algorithmic_flops"] / tc / 1e12, 1),
            "image_bit_identical_with_dense": culled[2]}
        r["culled"]["traffic"], r["culled"]["traffic_source"] = measured_traffic("splat_fwd_mfma_tile(culled)", w.N, w.B, w.R)
        r["culled"]["algorithmic_bytes"] = 4.0 * w.B * w.R * w.R + 16.0 * live * w.B * w.N * 4 + 8.0 * w.R
        # the compaction launch in front of it: reads every ray once per tile row … writes the lists
        r["culled"]["list_kernel_traffic"], r["culled"]["list_kernel_traffic_source"] = measured_traffic("cull_fwd_kernel", w.N, w.B, w.R)
    # backward at the same size: forward + backward kernels for given cotangents (no autograd graph), default
    # (culled) and dense, interleaved
    try:
        G = torch.randn((w.B, w.R, w.R), device=dev)
        ones = torch.ones((w.B, w.N, 3), device=dev)

        def fwd_bwd():
            return field.render_value_and_grad(suns_d, action, G, ones)

        def fwd_bwd_dense():
            ops.cull = False
            try:
                return field.render_value_and_grad(suns_d, action, G, ones)
            finally:
                ops.cull = True

        g_c, g_d = fwd_bwd()[2], fwd_bwd_dense()[2]
        tb = time_interleaved([fwd_bwd, fwd_bwd_dense], max(5, iters // 2))
        r["fwd_bwd"] = {"what": "HelioField.render_value_and_grad at config 4: geometry + splat forward, splat backward (two "
                                "MFMA passes) + geometry backward, for given cotangents; default (culled) and dense",
                        "ms": round(tb[0] * 1e3, 3), "dense_ms": round(tb[1] * 1e3, 3),
                        "frames_per_s": round(w.B / tb[0], 1),
                        "gradient_bit_identical_with_dense": bool(torch.equal(g_c.view(torch.int32), g_d.view(torch.int32)))}
        # the backward's footprint kernels alone (helio_splat_bwd: both contractions, one launch + the tail form's), dense and
        # with the lists — the backward's own roofline entry: 4·B·N·R² flops (two fused multiply-adds per (ray, pixel))
        trig_b, stride_b = field._select_trig(w.B)
        rays_b = ops.geometry_fwd(field.heliostat_positions, suns_d, action.reshape(w.B, w.N, 3).contiguous(), trig_b, stride_b, field._plane)[2]
        t_bd, t_bc = time_interleaved([lambda: ops.splat_bwd(rays_b, field._xs, field._ys, G, variant=0, cull=False),
                                       lambda: ops.splat_bwd(rays_b, field._xs, field._ys, G, variant=0)], max(5, iters // 2))
        bflops = 2.0 * r["algorithmic_flops"]
        r["fwd_bwd"]["splat_bwd"] = {"bound": "mfma", "kernel": "splat_bwd_mfma_both (variant %d)" % ops.render_bwd_choice(w.B, w.N, w.R),
                                     "dense_us": round(t_bd * 1e6, 1), "achieved": round(bflops / t_bd / 1e12, 2), "peak": F32_MFMA_PEAK_TF,
                                     "unit": "TFLOP/s", "frac": round(bflops / t_bd / 1e12 / F32_MFMA_PEAK_TF, 4),
                                     "algorithmic_flops": bflops, "with_lists_us": round(t_bc * 1e6, 1)}
        del G, ones, g_c, g_d, rays_b
    except Exception as e:  # noqa: BLE001
        r["fwd_bwd"] = {"error": repr(e)}
    # the opt-in split-bf16 kernel (HELIO_SPLAT_VARIANT=7) on the same rays, beside the exact-f32 default:
    # never the roofline entry above, reported so that its speed and its accuracy can be judged together
    try:
        trig, stride = field._select_trig(w.B)
        normals = action.reshape(w.B, w.N, 3).contiguous()
        _, _, rays = ops.geometry_fwd(field.heliostat_positions, suns_d, normals, trig, stride, field._plane)
        exact = ops.splat_fwd(rays, field._xs, field._ys, variant=0)
        split = torch.empty_like(exact)
        r["split_bf16_kernel"] = {
            "note": "opt-in: exact 3-way bf16 split of every f32 factor, 6 of the 9 partial products on "
                    "v_mfma_f32_32x32x16_bf16, f32 accumulation; dropped terms < 2^-23 of each product.  Variant 7 "
                    "sums in two levels (more accurate against fp64 than the exact-f32 kernel's one-level chain: "
                    "tools/accuracy_splat.py), variant 8 in one"}
        for v, name in ((7, "two_level"), (8, "one_level")):
            args = (w.B, w.N, w.R, rays.data_ptr(), field._xs.data_ptr(), field._ys.data_ptr(), split.data_ptr(), v,
                    None, 0, native._stream())
            t7 = time_kernel(lambda: ops.lib.helio_splat_fwd(*args), iters)
            rel = ((split - exact).abs() / exact.clamp_min(1e-6 * exact.max())).max().item()
            r["split_bf16_kernel"][name] = {
                "variant": v, "kernel_us": round(t7 * 1e6, 2),
                "speedup_vs_f32_mfma_kernel": round(r["kernel_us"] / (t7 * 1e6), 3),
                "f32_equivalent_TFLOPs": round(r["algorithmic_flops"] / t7 / 1e12, 1),
                "bf16_mfma_TFLOPs_issued": round(6 * r["algorithmic_flops"] / t7 / 1e12, 1),
                "max_rel_deviation_from_f32_kernel": float(f"{rel:.3e}")}
    except Exception as e:  # noqa: BLE001
        r["split_bf16_kernel"] = {"error": repr(e)}
    r["workload"] = w.name + f", span={w.span} m, sigma_scale={w.sigma_scale}, err={w.error_scale_mrad} mrad"
    r["sigma_px"] = sigma_pixels(field, suns_d, action)
    r["render_frames_per_s"] = round(w.B / el, 1)
    r["render_ms"] = round(el * 1e3, 3)
    r["render_calls_timed"] = iters
    r["timing"] = ("kernel_us (dense kernel), culled.splat_us and render_ms (field.render: geometry + culled splat) from ONE "
                   "interleaved loop, every call between its own HIP events")
    fwd_bytes = 4.0 * w.B * w.R * w.R + 32.0 * w.B * w.N + 12.0 * w.N + 12.0 * w.B
    r["render_hbm_GBs"] = round(fwd_bytes / el / 1e9, 1)
    r["render_hbm_frac"] = round(fwd_bytes / el / 1e9 / HBM_PEAK_GBS, 4)
    return r


if __name__ == "__main__":
    main()
