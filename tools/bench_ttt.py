#!/usr/bin/env python3
"""The reference's TTT sweep shape (run_experiments.py:31-56): batch_size=500, num_heliostats=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd.env import HelioEnv
dev = "cuda"
torch.manual_seed(0)
N, B, R = 1, 500, 128
hp = torch.rand(N, 3, device=dev) * 10 + 80; hp[:, 2] = 0
env = HelioEnv(hp, torch.tensor([0., -5., 0.], device=dev), (15., 15.), torch.tensor([0., 1., 0.], device=dev),
               sigma_scale=0.01, error_scale_mrad=2.0, resolution=R, batch_size=B, device=dev)
env.reset()
a = torch.nn.functional.normalize(env.ideal_normals + 0.001 * torch.randn_like(env.ideal_normals), dim=2).reshape(B, -1).requires_grad_(True)
def timeit(fn, n=200, repeats=5):
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
def fb():
    _, m, _ = env.step(a); m["dist"].backward(); a.grad = None
with torch.no_grad():
    print(f"N=1 B=500 R=128: env.step fwd {timeit(lambda: env.step(a)):.1f} us")
print(f"                 env.step + dist.backward {timeit(fb):.1f} us")

# the same iteration replayed from a HIP graph (doodle_amd/graphed.py)
from doodle_amd.graphed import GraphedEnvStep
base = env.ideal_normals.detach()
vec = torch.empty_like(base).uniform_(-1e-3, 1e-3)
g = GraphedEnvStep(env, like=vec, objective="dist", prepare=lambda v: torch.nn.functional.normalize(base + v, dim=2))
print(f"                 graphed normalize + env.step + dist.backward {timeit(lambda: g()):.1f} us")
v = vec.clone().requires_grad_(True)
def eager():
    _, m, _ = env.step(torch.nn.functional.normalize(base + v, dim=2)); m["dist"].backward(); v.grad = None
print(f"                 eager   normalize + env.step + dist.backward {timeit(eager):.1f} us")
