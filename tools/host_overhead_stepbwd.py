#!/usr/bin/env python3
"""Where the wall time of `env.step(a); metrics['dist'].backward()` goes at config 2: the two C calls
alone, the autograd.Function around them, and the whole Python surface."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd.env import HelioEnv
from doodle_amd.losses import StepConstants, env_step_fused
from doodle_amd import field as _field

dev = "cuda"
torch.manual_seed(0)
N, B, R = 50, 25, 128
hp = torch.rand(N, 3, device=dev) * 10 + 80; hp[:, 2] = 0
env = HelioEnv(hp, torch.tensor([0., -5., 0.], device=dev), (15., 15.), torch.tensor([0., 1., 0.], device=dev),
               sigma_scale=0.01, error_scale_mrad=90.0, resolution=R, batch_size=B, device=dev)
env.reset()
act = torch.nn.functional.normalize(env.ideal_normals + 0.003 * torch.randn_like(env.ideal_normals), dim=2).reshape(B, -1)
a = act.clone().requires_grad_(True)

def t(fn, n=1000, repeats=5):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: fn()
    best = 1e9
    for _ in range(repeats):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best

ops = _field._get_ops()
f = env.noisy_field
ideal, target, tx, _ = env._reference()
c = StepConstants(target, tx, env.distance_maps, ideal, f.heliostat_positions, env._tp3, env._tn3, 15.0, 15.0, False)
trig, stride = f._select_trig(B)
normals = act.view(B, N, 3).contiguous()
one = torch.ones((), device=dev)

def raw():
    r = ops.env_step_fwd(f.heliostat_positions, env.sun_pos, normals, trig, stride, f._plane, f._xs, f._ys, c)
    ops.env_step_bwd(f.heliostat_positions, env.sun_pos, normals, trig, stride, f._plane, r[3], f._xs, f._ys, r[0], c,
                     None, one, None, None, r[8], None, None)
print("two C calls (env_step_fwd + env_step_bwd), no autograd   %.1f us" % t(raw))

nr = normals.clone().requires_grad_(True)
def node():
    out = env_step_fused(f, env.sun_pos, nr, c)
    out[4].backward()
    nr.grad = None
print("the same through the autograd.Function + backward()      %.1f us" % t(node))

def step_only():
    env.step(a)
print("env.step(a) with grad, no backward                        %.1f us" % t(step_only))
env.check_finite = False
print("env.step(a) with grad, no backward, check_finite=False    %.1f us" % t(step_only))
env.check_finite = True

def full():
    _, m, _ = env.step(a)
    m["dist"].backward()
    a.grad = None
print("env.step(a) + dist.backward()                             %.1f us" % t(full))
