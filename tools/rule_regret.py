#!/usr/bin/env python3
"""Rule regret: how far the size rules' own choice (variant 0) is from the best forced variant, forward and backward.

For every (B, N, R) of tools/sweep_render.py's grid plus the neighbourhood of the sizes the reference runs
(R in {100, 128, 256} x N in {1, 8, 50, 96, 200} x B in {1, 4, 25, 60, 500}: train_with_env.py:227-241 is N=50, B=25,
R=128; run_experiments.py:31-56 is N=1, B=500) one helio_render_fwd / helio_render_bwd call is timed (HIP events, least
of three loops) with variant 0 and with every variant that exists at that size.  regret = t(auto) / t(best) - 1.
The split-bf16 kernels (forward 7 / 8, backward 5) are opt-in, never chosen by the rules, and do not count as "best".
Inputs: err 40 mrad, sigma_scale 0.02 — every ray lands on the image, so no list shortens anybody's work.
Times are DEVICE times: HIP events around an eager loop where a call takes longer than the host needs to issue it, a
HIP-graph replay of 20 back-to-back calls below that.  What the host pays per call (one launch or two) is not in them;
the last section times config 3's forward + backward through the Python surface for the two backward forms it concerns.

usage: rule_regret.py [quick] [err90] [sizes=B,N,R;B,N,R…] [out.txt]      (err90: err 90 mrad, sigma_scale 0.01 instead — the lists bite)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import build_field, make_action, time_kernel  # noqa: E402
from doodle_amd import native, synthetic  # noqa: E402

dev = torch.device("cuda")
ops = native.get_ops()
WALL_SIZES = {(25, 50, 128), (1, 50, 128), (4, 8, 128), (60, 32, 128), (4, 200, 100), (25, 8, 100)}
FWD = (1, 3, 4, 5, 6, 9, 10, 11, 12, 13, 14, 15, 16, 17)
FWD_OPT_IN = (7, 8)
BWD = (1, 2, 4, 6, 7, 8, 9, 10, 11, 12)
BWD_OPT_IN = (5,)


def grid(quick):
    g = [(B, N, R) for R in (64, 128, 256, 512) for N in (50, 200, 1000, 5000) for B in (4, 32, 256)
         if B * N * R * R <= 3e11]
    g += [(B, N, R) for R in (100, 128, 256) for N in (1, 8, 50, 96, 200) for B in (1, 4, 25, 60, 500)]
    g += [(B, N, R) for R in (128, 256) for N in (2, 4, 16, 32) for B in (4, 60, 500)]      # around the few-ray rules
    seen, out = set(), []
    for p in g:
        if p not in seen:
            seen.add(p)
            out.append(p)
    return out[::5] if quick else out


def timed(fn, flops):
    """Device time per call (s), or None where the variant does not exist.  Calls shorter than ≈100 µs are timed as a
    HIP-graph replay of 20 back-to-back calls (least of five replays): an eager Python loop cannot issue faster than
    ≈6 µs per call, which is ALL a loop of the small sizes would show, for every variant alike."""
    iters = max(5, min(200, int(1e11 / max(flops, 1.0))))
    try:
        fn()
        torch.cuda.synchronize()
    except RuntimeError:
        return None                                   # this variant does not exist at this size
    t = time_kernel(fn, iters, warm=3, repeats=3)
    if t > 100e-6:
        return t
    K = 20
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            fn()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / K
        best = dt if best is None else min(best, dt)
    del g
    return best


def main():
    quick = "quick" in sys.argv[1:]
    culled = "err90" in sys.argv[1:]           # the reference's default error scale and sigma: the lists shorten the list-taking kernels' work
    sigma, err = (0.01, 90.0) if culled else (0.02, 40.0)
    custom = [a for a in sys.argv[1:] if a.startswith("sizes=")]       # sizes=B,N,R;B,N,R…: these sizes instead of the grid
    outs = [a for a in sys.argv[1:] if a not in ("quick", "err90") and not a.startswith("sizes=")]
    lines, rows, wall_jobs = [], [], []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    emit(f"# err {err} mrad, sigma_scale {sigma}")
    emit(f"{'B':>4} {'N':>5} {'R':>4} | {'fwd auto':>9} {'=v':>3} {'best':>9} {'v':>3} {'regret':>7} | "
         f"{'bwd auto':>9} {'=v':>3} {'best':>9} {'v':>3} {'regret':>7}")
    sizes = [tuple(int(x) for x in t.split(",")) for t in custom[0][6:].split(";")] if custom else grid(quick)
    for B, N, R in sizes:
        w = synthetic.Workload("s", N=N, B=B, R=R, sigma_scale=sigma, error_scale_mrad=err, span=30.0 if N > 100 else 10.0)
        helios, suns, errs, noise = synthetic.make_inputs(w, 0)
        f = build_field(w, helios, errs, dev)
        suns_d = suns.to(dev)
        act = make_action(f, suns_d, noise)
        trig, stride = f._select_trig(B)
        normals = act.reshape(B, N, 3).contiguous()
        G = torch.randn(B, R, R, device=dev)
        flops = 2.0 * B * N * R * R
        hp, pl, xs, ys = f.heliostat_positions, f._plane, f._xs, f._ys
        with torch.no_grad():
            rays = ops.render_fwd(hp, suns_d, normals, trig, stride, pl, xs, ys)[3]
            tf, tb = {}, {}
            for v in (0,) + FWD + FWD_OPT_IN:
                if 14 <= v <= 17 and N < 128 * (2 << (v - 14)):        # a split heliostat sum wants parts of >= 128 rays
                    tf[v] = None
                    continue
                tf[v] = timed(lambda: ops.render_fwd(hp, suns_d, normals, trig, stride, pl, xs, ys, rays=rays, variant=v), flops)
            # (the rules' choice once more, LAST: the first timing of a size runs on clocks that dropped while the host
            # built the field — 8–13 % at the 150 µs sizes; the lesser of the two counts)
            v = 0
            tf[0] = min(tf[0], timed(lambda: ops.render_fwd(hp, suns_d, normals, trig, stride, pl, xs, ys, rays=rays, variant=v), flops))
            ops.render_fwd(hp, suns_d, normals, trig, stride, pl, xs, ys, rays=rays)
            for v in (0,) + BWD + BWD_OPT_IN:
                tb[v] = timed(lambda: ops.render_bwd(hp, suns_d, normals, trig, stride, pl, rays, xs, ys, G, None, None, variant=v), 2 * flops)
            v = 0
            tb[0] = min(tb[0], timed(lambda: ops.render_bwd(hp, suns_d, normals, trig, stride, pl, rays, xs, ys, G, None, None, variant=v), 2 * flops))
        bf = min((v for v in FWD if tf[v] is not None), key=lambda v: tf[v])
        bb = min((v for v in BWD if tb[v] is not None), key=lambda v: tb[v])
        rf, rb = tf[0] / tf[bf] - 1.0, tb[0] / tb[bb] - 1.0
        cf, cb = ops.render_choice(B, N, R), ops.render_bwd_choice(B, N, R)
        # the latency-bound corner where the backward's rule follows the WALL clock of a forward + backward through the
        # Python surface (one launch less), not the device time (splat_bwd.hip, render_bwd_is_fused (a)): marked, listed apart
        host = cb == 8 and R <= 128 and B * ((N + 31) // 32) <= 64
        emit(f"{B:4d} {N:5d} {R:4d} | {tf[0] * 1e6:9.1f} {cf:3d} {tf[bf] * 1e6:9.1f} {bf:3d} {rf * 100:6.1f}% | "
             f"{tb[0] * 1e6:9.1f} {cb:3d} {tb[bb] * 1e6:9.1f} {bb:3d} {rb * 100:6.1f}%{' *' if host else ''}")
        rows.append((B, N, R, "fwd", cf, bf, tf[0], tf[bf], rf, {v: t for v, t in tf.items() if t is not None}, False))
        rows.append((B, N, R, "bwd", cb, bb, tb[0], tb[bb], rb, {v: t for v, t in tb.items() if t is not None}, host))
        if host and (B, N, R) in WALL_SIZES:
            wall_jobs.append((B, N, R, f, suns_d, act))
            continue            # (the field is kept for the wall-clock section)
        del f, G, rays
        torch.cuda.empty_cache()
    emit("")
    dev_rows = [r for r in rows if not r[10]]
    emit("worst ten on the device clock (regret of the rules' choice against the best forced variant; rows marked * apart):")
    for B, N, R, which, c, best, t0, tbest, reg, allv, _ in sorted(dev_rows, key=lambda r: -r[8])[:10]:
        every = " ".join(f"v{v}={t * 1e6:.1f}" for v, t in sorted(allv.items()))
        emit(f"  {which} B={B} N={N} R={R}: auto (= v{c}) {t0 * 1e6:.1f} us, best v{best} {tbest * 1e6:.1f} us, regret {reg * 100:.1f}%   [{every}]")
    emit(f"max regret {max(r[8] for r in dev_rows) * 100:.1f}% over {len(dev_rows)} (size, direction) pairs; "
         f"over 10 %: {sum(1 for r in dev_rows if r[8] > 0.10)}")
    star = [r for r in rows if r[10]]
    if star:
        emit(f"rows marked * ({len(star)}: the single-launch backward where it is at most 64 workgroups, R <= 128): on the device clock the two "
             f"launches are {min(r[8] for r in star) * 100:.0f}–{max(r[8] for r in star) * 100:.0f} % shorter; the rule follows the wall clock of a "
             "forward + backward through the Python surface, where a launch less is worth more:")
        import time
        for B, N, R, f, suns_d, act in wall_jobs:
            G = torch.randn(B, R, R, device=dev)
            H = torch.ones(B, N, 3, device=dev)
            res = {}
            for v in (0, 10, 11):
                ops.bwd_variant = v
                try:
                    for _ in range(300):
                        f.render_value_and_grad(suns_d, act, G, H)
                    torch.cuda.synchronize()
                    best = None
                    for _ in range(5):
                        t0 = time.perf_counter()
                        for _ in range(1000):
                            f.render_value_and_grad(suns_d, act, G, H)
                        torch.cuda.synchronize()
                        dt = (time.perf_counter() - t0) / 1000
                        best = dt if best is None else min(best, dt)
                finally:
                    ops.bwd_variant = 0
                res[v] = best
            emit(f"  B={B} N={N} R={R}: HelioField.render_value_and_grad, wall clock per call (best of 5 loops of 1000): by rule (v8) "
                 f"{res[0] * 1e6:.2f} us, two launches v10 {res[10] * 1e6:.2f} / v11 {res[11] * 1e6:.2f} us")
    if outs:
        with open(outs[0], "w") as fh:
            fh.write("\n".join(lines) + "\n")
        import json
        with open(os.path.splitext(outs[0])[0] + ".json", "w") as fh:      # every variant's time, for deriving rules offline
            json.dump([{"B": r[0], "N": r[1], "R": r[2], "dir": r[3], "auto": r[4], "wall_clock_rule": r[10],
                        "us": {str(v): round(t * 1e6, 2) for v, t in r[9].items()}} for r in rows], fh)


if __name__ == "__main__":
    main()
