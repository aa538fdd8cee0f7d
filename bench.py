#!/usr/bin/env python3
"""bench.py — HelioField.render frames/s on MI355X (driver contract, see README).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg4] [--no-cpu]

A "step" is one ``HelioField.render`` forward over one batch of B synthetic sun
positions (B frames), inputs resident in HBM.  With N>1 (launched by
``torch.distributed.run``, one rank per GPU) every rank renders its own B-row shard of
a global batch of N·B suns (weak scaling).  The suns are independent, so the timed loop has
no data-path collective: the images stay on the rank that rendered them, which is how a
data-parallel training job consumes them.  The path's one collective — the RCCL all-gather
of image shards, for a single consumer that wants the whole batch — is timed right after, on
the same shards, and reported beside `value` as `with_all_gather_every_step` (at config 2 it
is interconnect-bound by construction: DESIGN.md §5); `--gather-every-step` makes it the
timed loop instead.  The config-5 shard leg always renders AND gathers (the configuration
BASELINE names with the all-gather).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline        the dominant kernel of the run by GPU time: the splat forward at BASELINE
                  config 4 (N=2000, B=512, R=512), timed live with HIP events on the launch
                  stream (f32 MFMA roofline; the HBM view is given beside it)
  roofline_bench_workload
                  the (launch-latency-bound) kernel of the timed config-2 loop, same method
  cpu_baseline    the oracle (oracle/torch_oracle.py, a CPU PyTorch restatement of the
                  reference that is bit-identical with it) timed on this host's cores
"""
from __future__ import annotations

import argparse
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


def time_kernel(fn, iters, warm=3):
    """Average duration (s) of ``fn`` — one kernel launch — over ``iters`` back-to-back
    launches, bracketed by HIP events on the current (launch) stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def splat_roofline(field, suns, action, iters, variant=None):
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
                rays.data_ptr(), image.data_ptr(), var, st)
        t = time_kernel(lambda: lib.helio_render_fwd(*args), iters)
        bytes_alg = 4.0 * B * R * R + 32.0 * B * N + 16.0 * B * N + 12.0 * N + 12.0 * B + 8.0 * R
    else:
        kernel = "splat_fwd"
        args = (B, N, R, rays.data_ptr(), field._xs.data_ptr(), field._ys.data_ptr(), image.data_ptr(), var, st)
        t = time_kernel(lambda: lib.helio_splat_fwd(*args), iters)
        bytes_alg = 4.0 * B * R * R + 16.0 * B * N + 8.0 * R   # image store + ray parameters + xs/ys
    flops = 2.0 * B * N * R * R                       # one FMA per (ray, pixel)
    return {
        "bound": "mfma", "kernel": kernel, "achieved": round(flops / t / 1e12, 3),
        "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(flops / t / 1e12 / F32_MFMA_PEAK_TF, 4),
        "traffic": measured_traffic(kernel if fused else "splat_fwd_mfma_tile", N, B, R),
        "kernel_us": round(t * 1e6, 2),
        "hbm_achieved_GBs": round(bytes_alg / t / 1e9, 1), "hbm_frac": round(bytes_alg / t / 1e9 / HBM_PEAK_GBS, 4),
        "algorithmic_bytes": bytes_alg, "algorithmic_flops": flops,
    }


def measured_traffic(kernel, N, B, R):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_traffic.json), or None
    when no pass was taken for this kernel at this size (bench.py cannot profile itself)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            return json.load(f).get(f"{kernel}@N={N},B={B},R={R}", {}).get("hbm_bytes")
    except OSError:
        return None


def cpu_baseline(w, seed, budget_s=12.0):
    """The oracle timed on this host: forward render of the bench workload."""
    from oracle import torch_oracle as to
    helios, suns, errs, noise = synthetic.make_inputs(w, seed)
    sc = to.Scene.build(helios, synthetic.TARGET_POSITION, synthetic.TARGET_AREA, synthetic.TARGET_NORMAL,
                        w.R, w.sigma_scale)
    ideal = to.ideal_normals(helios, sc.target_position, suns)
    a = ideal + noise
    a = (a / a.norm(dim=2, keepdim=True)).reshape(w.B, -1)
    with torch.no_grad():
        to.render(sc, suns, a, errs)                     # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            to.render(sc, suns, a, errs)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 40:
                break
    return {"value": round(w.B * n / el, 2), "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "host_cpus": os.cpu_count(),
            "sample": f"{n} forward renders of {w.name} (all {w.B} suns, whole workload), {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 20000 at the microsecond-scale configs — a 2000-step sample is 13 ms "
                         "and reads 5-10 %% slower from host jitter — 50 at cfg4, 10 at cfg5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps first (default: a tenth of --steps)")
    ap.add_argument("--workload", default="cfg2", choices=sorted(synthetic.CONFIGS))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-large", action="store_true", help="skip the config-4 roofline leg")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary legs (fwd+bwd, env.step, config-5 shard): used for profiling runs "
                         "so that every kernel in the trace belongs to one workload")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "fwdbwd"])
    ap.add_argument("--overlap", action="store_true",
                    help="run the all-gather on a side stream (pays for steps of milliseconds — the config-5 "
                         "shard leg uses it; at config 2 the cross-stream events cost more than they hide: "
                         "35 vs 14 µs per step measured with one rank)")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="all-gather the images inside the timed loop (default: images stay on their rank; the "
                         "gathered loop is timed afterwards and reported as with_all_gather_every_step)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and all-gather even with one rank (testing)")
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
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)          # only matters for a bare --force-dist run
        dist.init_process_group("nccl", device_id=dev)

    w = synthetic.CONFIGS[args.workload]
    helios, suns, errs, noise = synthetic.make_inputs(w, args.seed, b_offset=rank * w.B, b_count=w.B)
    field = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    action = make_action(field, suns_d, noise)
    gather, gathered, gather_error = None, None, None
    if dist is not None:
        # the timed loop does not need the collective; if the communicator cannot be created the
        # gathered legs are skipped and said so, the headline is still measured
        try:
            from doodle_amd.comm import ImageGather
            gather = ImageGather()             # RCCL all-gather (libhelio_comm.so)
            gathered = [torch.empty((world * w.B, w.R, w.R), dtype=torch.float32, device=dev) for _ in range(2)]
        except Exception as e:  # noqa: BLE001
            gather, gather_error = None, repr(e)
        flag = torch.tensor([0 if gather is not None else 1], device=dev)
        dist.all_reduce(flag)                  # all ranks gather, or none does
        if int(flag.item()) != 0 and gather is not None:
            gather.close()
            gather, gather_error = None, "another rank could not create the communicator"
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

    gather_now = [bool(args.gather_every_step)]
    for _ in range(args.warmup):
        step()

    def fence():
        if gather is not None:
            gather.wait()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    if args.mode == "fwd" and not gather_now[0]:
        # the timed loop proper: K calls of HelioField.render and nothing else (config 2 is ≈5.3 µs of
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
                fence()
                el = time.perf_counter() - t0
        finally:
            if gc_was:
                gc.enable()
    else:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # the same shards with the other treatment of the images (gathered every step / left on their
    # rank): at config 2 the gathered loop is bound by delivering (N-1) x 1.64 MB to every rank per
    # step, DESIGN.md §5 — reported beside `value`, never instead of it
    el_other = None
    if gather is not None and (world > 1 or args.force_dist):
        gather_now[0] = not gather_now[0]
        for _ in range(min(args.warmup, 50)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_other = float(t.item())
        gather_now[0] = not gather_now[0]

    # secondary weak-scaling point on EVERY rank (collective): a config-5-like shard, where
    # compute (ms) dominates the all-gather — see DESIGN.md §5
    shard = None
    if not args.no_large and not args.no_extras:
        try:
            shard = cfg5_shard_leg(dev, rank, world, gather, dist, args.seed)
        except Exception as e:  # noqa: BLE001
            shard = {"error": repr(e)}

    if rank == 0:
        frames = world * w.B * args.steps
        out = {
            "metric": "HelioField.render frames/sec", "value": round(frames / el, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{w.name} {'forward' if args.mode == 'fwd' else 'forward+backward'} "
                                   f"HelioField.render, sigma_scale={w.sigma_scale}, err={w.error_scale_mrad} mrad, "
                                   f"{w.B} suns per GPU",
                       "global_batch": world * w.B, "parallelism": f"sun-batch sharded x{world}"
                       + ((f", RCCL all-gather of images every step ({gather.transport} transport, "
                           f"{'side stream' if args.overlap else 'stream-ordered'})") if (gather is not None and args.gather_every_step)
                          else (", no data-path collective (images stay on the rank that rendered them)" if world > 1 else ""))},
        }
        if el_other is not None:
            key = "without_all_gather" if args.gather_every_step else "with_all_gather_every_step"
            out[key] = {"frames_per_s": round(frames / el_other, 1), "ms_per_step": round(el_other / args.steps * 1e3, 5),
                        "note": ("same shards, images left on the rank that rendered them" if args.gather_every_step else
                                 f"same shards, every image delivered to every rank each step ({gather.transport} transport): "
                                 "interconnect-bound at this frame size, DESIGN.md §5")}
        if gather_error is not None:
            out["all_gather_error"] = gather_error
        if shard is not None:
            out["weak_scaling_config5_shard"] = shard
        iters = 200 if w.B * w.N * w.R * w.R < 1e10 else 10
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
            if not args.no_cpu:
                out["cpu_baseline"] = cpu_baseline(w, args.seed)
                out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
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
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:      # a fresh Python path runs slow for its first ~0.1 s
            fn()
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

    t_fb = wall(fwdbwd, 300)
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
    return {"render_fwd_bwd_frames_per_s": round(w.B / t_fb, 1), "render_fwd_bwd_us": round(t_fb * 1e6, 1),
            "env_step_fwd_frames_per_s": round(w.B / t_step, 1), "env_step_fwd_us": round(t_step * 1e6, 1),
            "env_step_fwd_bwd_us": round(t_sb * 1e6, 1),
            "note": "config 3 = render + autograd.grad of (img*G).sum()+actual.sum(); env.step = render + loss "
                    "block (2 launches) + NaN/Inf check (one wait on a pinned host record); fwd_bwd adds "
                    "metrics['dist'].backward(); best of 3 wall-clock loops through the Python surface"}


def cfg5_shard_leg(dev, rank, world, gather, dist, seed, b_local=256, steps=8):
    """Every rank renders its own ``b_local``-sun shard of BASELINE config 5 (N=5000, R=256) and the
    ranks all-gather the images; whole-job frames/s.  Compute per step ≈ ms, so this is the
    regime in which the sun-batch sharding scales; config 2 (the headline) is gather-bound."""
    w5 = synthetic.CONFIGS["cfg5"]
    w = synthetic.Workload(w5.name, w5.N, b_local, w5.R, w5.sigma_scale, w5.error_scale_mrad, w5.span)
    helios, suns, errs, noise = synthetic.make_inputs(w, seed, b_offset=rank * b_local, b_count=b_local)
    field = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    action = make_action(field, suns_d, noise)
    out = torch.empty((world * b_local, w.R, w.R), dtype=torch.float32, device=dev) if gather is not None else None

    def step():
        with torch.no_grad():
            img, _ = field.render(suns_d, action, None)
        if gather is not None:
            # side stream: the gather of this step's images (67 MB per rank) overlaps the next render
            gather.gather(img, out[step.k & 1], overlap=gather.transport == "rccl")
            step.k += 1

    step.k = 0
    if gather is not None:
        out = [out, torch.empty_like(out)]
    for _ in range(3):
        step()
    if gather is not None:
        gather.wait()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if gather is not None:
        gather.wait()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return {"workload": f"N={w.N}, R={w.R}, {b_local} suns per GPU", "n_gpus": world,
            "frames_per_s": round(world * b_local * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 3),
            "gathered_MB_per_rank_and_step": round(world * b_local * w.R * w.R * 4 / 1e6, 1)}


def large_leg(dev, seed):
    """Splat-forward roofline and whole-render rate at BASELINE config 4."""
    w = synthetic.CONFIGS["cfg4"]
    helios, suns, errs, noise = synthetic.make_inputs(w, seed)
    field = build_field(w, helios, errs, dev)
    suns_d = suns.to(dev)
    action = make_action(field, suns_d, noise)
    r = splat_roofline(field, suns_d, action, iters=5)
    with torch.no_grad():
        for _ in range(2):
            field.render(suns_d, action, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            field.render(suns_d, action, None)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 5
    # the opt-in split-bf16 kernel (HELIO_SPLAT_VARIANT=7) on the same rays, beside the exact-f32 default:
    # never the roofline entry above, reported so that its speed and its accuracy can be judged together
    try:
        ops = native.get_ops()
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
                    native._stream())
            t7 = time_kernel(lambda: ops.lib.helio_splat_fwd(*args), 5)
            rel = ((split - exact).abs() / exact.clamp_min(1e-6 * exact.max())).max().item()
            r["split_bf16_kernel"][name] = {
                "variant": v, "kernel_us": round(t7 * 1e6, 2),
                "speedup_vs_f32_mfma_kernel": round(r["kernel_us"] / (t7 * 1e6), 3),
                "f32_equivalent_TFLOPs": round(r["algorithmic_flops"] / t7 / 1e12, 1),
                "bf16_mfma_TFLOPs_issued": round(6 * r["algorithmic_flops"] / t7 / 1e12, 1),
                "max_rel_deviation_from_f32_kernel": float(f"{rel:.3e}")}
    except Exception as e:  # noqa: BLE001
        r["split_bf16_kernel"] = {"error": repr(e)}
    r["workload"] = w.name + f", span={w.span} m, sigma_scale={w.sigma_scale}"
    r["render_frames_per_s"] = round(w.B / el, 1)
    r["render_ms"] = round(el * 1e3, 3)
    fwd_bytes = 4.0 * w.B * w.R * w.R + 32.0 * w.B * w.N + 12.0 * w.N + 12.0 * w.B
    r["render_hbm_GBs"] = round(fwd_bytes / el / 1e9, 1)
    r["render_hbm_frac"] = round(fwd_bytes / el / 1e9 / HBM_PEAK_GBS, 4)
    return r


if __name__ == "__main__":
    main()
