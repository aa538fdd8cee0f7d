#!/usr/bin/env python3
"""HelioEnv.step timing on MI355X at the README/training configuration (N=50, B=25, R=128)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd.env import HelioEnv

dev = "cuda"
torch.manual_seed(0)
N, B, R = 50, 25, 128
hp = torch.rand(N, 3, device=dev) * 10 + 80; hp[:, 2] = 0
env = HelioEnv(hp, torch.tensor([0., -5., 0.], device=dev), (15., 15.), torch.tensor([0., 1., 0.], device=dev),
               sigma_scale=0.01, error_scale_mrad=90.0, resolution=R, batch_size=B, device=dev)
obs = env.reset()
act = (env.ideal_normals + 0.003 * torch.randn_like(env.ideal_normals))
act = torch.nn.functional.normalize(act, dim=2).reshape(B, -1)

def timeit(fn, n=300, repeats=5):
    """best of `repeats` timed loops (the GPU boxes' host cores are shared: single loops vary by ±20 %)"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: fn()          # host warm-up: the first ~0.1 s of a loop runs slow
    best = float("inf")
    for _ in range(repeats):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best

with torch.no_grad():
    t_fwd = timeit(lambda: env.step(act))
env.check_finite = "deferred"
with torch.no_grad():
    t_def = timeit(lambda: env.step(act))
    env.finish_checks()
env.check_finite = False
with torch.no_grad():
    t_off = timeit(lambda: env.step(act))
env.check_finite = True
a = act.clone().requires_grad_(True)
def fb(key):
    _, m, _ = env.step(a)
    m[key].backward()
    a.grad = None
timeit(lambda: fb("dist"), n=100, repeats=3)     # the FIRST differentiating loop of a process runs 1.5-2x slow for ~0.5 s
t_align = timeit(lambda: fb("alignment_loss"))    # (autograd's device thread, allocator pools) whichever metric it uses: untimed
t_dist = timeit(lambda: fb("dist"))
if "--ab" in sys.argv:      # A/B, interleaved: helio_env_step_bwd vs the composed backward (step_losses_bwd + render_bwd + add)
    from doodle_amd import field as _field
    ops = _field._get_ops()
    for rnd in range(3):
        one = (timeit(lambda: fb("alignment_loss")), timeit(lambda: fb("dist")))
        ops.env_step_bwd = None
        two = (timeit(lambda: fb("alignment_loss")), timeit(lambda: fb("dist")))
        del ops.env_step_bwd
        print(f"round {rnd}: one call {one[0]:7.1f} / {one[1]:7.1f} us   composed {two[0]:7.1f} / {two[1]:7.1f} us   (alignment / dist)")
print(f"env.step forward-only, finite check deferred by one step {t_def:8.1f} us | no check {t_off:8.1f} us")
print(f"env.step forward-only {t_fwd:8.1f} us = {B/t_fwd*1e6:10.0f} frames/s | step+backward(alignment) {t_align:8.1f} us | step+backward(dist) {t_dist:8.1f} us")
from doodle_amd.graphed import GraphedEnvStep
gs = GraphedEnvStep(env, like=act.reshape(B, -1, 3), objective="dist")
print(f"graph replay of step + dist gradient (doodle_amd/graphed.py) {timeit(lambda: gs()):8.1f} us")
with torch.no_grad():
    print(f"reset() {timeit(lambda: env.reset(), 100):8.1f} us")
if "--profile" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        with torch.no_grad():
            for _ in range(50): env.step(act)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
