#!/usr/bin/env python3
"""Why does the SAME config-2 workload read lower late in bench.py's process than in its headline loop?

BENCH_r03: `value` 5.48 M frames/s, `value_by_seed_and_sigma["sigma_scale=0.01"]["0"]` 4.17 M — same seed, same
sigma, same K = 20 loop, one process.  This script replays the headline sample (K bare field.render calls + synchronize,
median and best of 30 samples after a burst preheat) in a fresh process and again after each thing bench.py does between
the two readings, one at a time, so that the step that moves the figure shows:
  a second field (fresh tables, a new compiled context) / a side stream that has run work / a HIP-graph capture and
  replay / 20 GB allocated and freed through the caching allocator (+ empty_cache) / the config-4 forward (large
  launches, device scratch) / HelioEnv construction and steps (its streams, its pinned completion record) /
  the affinity widened to the NUMA node and narrowed again.
usage: headline_drift.py [out.txt]

FOUND (two runs, two boxes: profiles/r04_d_headline_drift.txt, r04_f_headline_drift.txt): the loop has two modes — 4.35 µs
per step and 5.3–6.5 µs (device period of a long loop 3.72 against 4.1–4.2 µs: the dispatch is slower, not the host's issue)
— and none of the listed state decides between them: the slow mode set in at different points in the two runs (with a side
stream and with 20 GB cached in one; after config-4-sized launches in the other, where the stream and the pool were neutral)
and ended by itself or when the threads' affinity was reset.  bench.py's late sweep now resets it before sampling."""
import os, sys, time, gc, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import affinity, native, synthetic
from bench import build_field, make_action, preheat

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.empty(1, device=dev)
numa = affinity.bind_to_gpu_ccd(0)
lines = []
def emit(s):
    print(s, flush=True); lines.append(s)

K = 20
def make(seed=0, sigma=0.01):
    w0 = synthetic.CONFIGS["cfg2"]
    w = synthetic.Workload(w0.name, w0.N, w0.B, w0.R, sigma, w0.error_scale_mrad, w0.span)
    helios, suns, errs, noise = synthetic.make_inputs(w, seed)
    f = build_field(w, helios, errs, dev)
    s = suns.to(dev)
    return w, f, s, make_action(f, s, noise)

def sample(f, s, a, what):
    with torch.no_grad():
        preheat(lambda: f.render(s, a, None), 0.3, burst=K, fence=torch.cuda.synchronize)
        gc.disable()
        ts = []
        try:
            render = f.render
            for _ in range(30):
                t0 = time.perf_counter()
                for _ in range(K):
                    render(s, a, None)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        finally:
            gc.enable()
    med, best = statistics.median(ts), min(ts)
    # … and where the time goes: the DEVICE period of the same call in a long back-to-back loop (HIP events, 3000 calls:
    # the GPU never waits for the host there) and the host's own cost of issuing it (wall clock of 3000 calls, no fence)
    with torch.no_grad():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(3000):
            f.render(s, a, None)
        e1.record()
        issue = (time.perf_counter() - t0) / 3000
        torch.cuda.synchronize()
        period = e0.elapsed_time(e1) * 1e-3 / 3000
    emit(f"{what:<62s} median {25 * K / med / 1e6:5.2f} M frames/s ({med / K * 1e6:5.2f} us/step)   best {25 * K / best / 1e6:5.2f} M"
         f"   | long loop: device period {period * 1e6:5.2f} us, host issue {issue * 1e6:5.2f} us per call")

w, f, s, a = make()
sample(f, s, a, "fresh process, first field")
sample(f, s, a, "the same again")
w2, f2, s2, a2 = make(seed=1)
sample(f2, s2, a2, "a second field (seed 1)")
sample(f, s, a, "back on the first field")

side = torch.cuda.Stream()
with torch.cuda.stream(side):
    x = torch.randn(1 << 20, device=dev); y = (x * 2).sum()
torch.cuda.synchronize()
sample(f, s, a, "after a side stream has run two kernels")
del side
sample(f, s, a, "… and the side stream is released")

g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    z = x * 3
torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
with torch.cuda.graph(g):
    z = x * 3
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
sample(f, s, a, "after a HIP-graph capture + 10 replays")
del g, st
sample(f, s, a, "… and the graph is released")

big = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(20)]
torch.cuda.synchronize(); del big
sample(f, s, a, "after 20 GB allocated and freed (cached by the allocator)")
torch.cuda.empty_cache()
sample(f, s, a, "… after torch.cuda.empty_cache()")

w4 = synthetic.CONFIGS["cfg4"]
h4, s4, e4, n4 = synthetic.make_inputs(synthetic.Workload(w4.name, w4.N, 64, w4.R, w4.sigma_scale, w4.error_scale_mrad, w4.span), 0)
f4 = build_field(synthetic.Workload(w4.name, w4.N, 64, w4.R, w4.sigma_scale, w4.error_scale_mrad, w4.span), h4, e4, dev)
s4d = s4.to(dev); a4 = make_action(f4, s4d, n4)
with torch.no_grad():
    for _ in range(5):
        f4.render(s4d, a4, None)
torch.cuda.synchronize()
sample(f, s, a, "after config-4-sized renders (64 suns: large launches, scratch)")
del f4, s4d, a4
torch.cuda.empty_cache()

from doodle_amd.env import HelioEnv
env = HelioEnv(f.heliostat_positions, torch.tensor([0., -5., 0.], device=dev), (15., 15.), torch.tensor([0., 1., 0.], device=dev),
               sigma_scale=0.01, error_scale_mrad=90.0, resolution=128, batch_size=25, device=dev, new_errors_every_reset=False)
env.reset()
with torch.no_grad():
    for _ in range(200):
        env.step(a)
torch.cuda.synchronize()
sample(f, s, a, "after HelioEnv construction + 200 steps (pinned completion record)")
ar = a.clone().requires_grad_(True)
for _ in range(100):
    _, m, _ = env.step(ar)
    m["dist"].backward()
torch.cuda.synchronize()
sample(f, s, a, "after 100 x env.step + dist.backward() (autograd engine thread)")
del env
if numa is not None:
    affinity.widen_to_node(0)
    sample(f, s, a, "with the affinity widened to the NUMA node")
    affinity.bind_to_gpu_ccd(0)
    sample(f, s, a, "… and narrowed to one CCD again")
w3, f3, s3, a3 = make(seed=0)
sample(f3, s3, a3, "a FRESH field of the same workload now (what value_sweep times)")
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
