#!/usr/bin/env python3
"""The reference README's quick start (README.md:61-95) on the MI355X render path.

    python examples/quickstart.py            # needs a gfx950 device and a built libhelio.so
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from doodle_amd import HelioEnv, HelioField

dev = "cuda"
torch.manual_seed(0)
N, B, R = 50, 25, 128
helios = torch.rand(N, 3, device=dev) * 10 + 80
helios[:, 2] = 0
target, normal, area = torch.tensor([0.0, -5.0, 0.0], device=dev), torch.tensor([0.0, 1.0, 0.0], device=dev), (15.0, 15.0)

# --- the optics core on its own -------------------------------------------------------------
field = HelioField(helios, target, area, normal, error_scale_mrad=2.0, sigma_scale=0.05, resolution=R, device=dev)
sun = torch.tensor([5000.0, 6000.0, 11000.0], device=dev)
field.init_actions(sun)
img, actual = field.render(sun, field.initial_action, field.calculate_ideal_normals(sun))
print("single sun :", tuple(img.shape), tuple(actual.shape), f"peak flux {img.max().item():.3f}")

# --- the Gym-style environment ----------------------------------------------------------------
env = HelioEnv(helios, target, area, normal, sigma_scale=0.05, error_scale_mrad=20.0, resolution=R, batch_size=B,
               device=dev)
obs = env.reset()
action = env.ideal_normals.reshape(B, -1).clone().requires_grad_(True)
obs, metrics, monitor = env.step(action)
metrics["dist"].backward()
print("env.step   :", {k: round(v.item(), 4) for k, v in metrics.items()}, "| grad norm", action.grad.norm().item())
