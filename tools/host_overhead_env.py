#!/usr/bin/env python3
"""Host-time breakdown of HelioEnv.step (forward, no grad) at config 2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd.env import HelioEnv
from doodle_amd.losses import StepConstants, step_losses

dev = "cuda"
torch.manual_seed(0)
N, B, R = 50, 25, 128
hp = torch.rand(N, 3, device=dev) * 10 + 80; hp[:, 2] = 0
env = HelioEnv(hp, torch.tensor([0., -5., 0.], device=dev), (15., 15.), torch.tensor([0., 1., 0.], device=dev),
               sigma_scale=0.01, error_scale_mrad=90.0, resolution=R, batch_size=B, device=dev)
env.reset()
act = torch.nn.functional.normalize(env.ideal_normals + 0.003 * torch.randn_like(env.ideal_normals), dim=2).reshape(B, -1)

def t(fn, n=3000, sync=False):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

with torch.no_grad():
    print("step (check_finite=True)   %.1f us" % t(lambda: env.step(act)))
    env.check_finite = False
    print("step (check_finite=False)  %.1f us" % t(lambda: env.step(act)))
    env.check_finite = True
    ideal, target, tx, _ = env._reference()
    print("_reference() cached        %.2f" % t(lambda: env._reference()))
    print("render monitor=True        %.2f" % t(lambda: env.noisy_field.render(env.sun_pos, act, ideal, monitor=True)))
    print("cat aux                    %.2f" % t(lambda: torch.cat([env.sun_pos.detach(), act.flatten(1)], dim=1)))
    img, actual, refl = env.noisy_field.render(env.sun_pos, act, ideal, monitor=True)
    normals = act.view(B, -1, 3)
    mk = lambda: StepConstants(target, tx, env.distance_maps, ideal, env.noisy_field.heliostat_positions, env._tp3, env._tn3, 15.0, 15.0, False)
    print("StepConstants()            %.2f" % t(mk))
    c = mk()
    print("step_losses                %.2f" % t(lambda: step_losses(img, actual, normals, c)))
    out = step_losses(img, actual, normals, c)
    print("bool(flag) after sync      %.2f" % t(lambda: bool(out[7])))
    from doodle_amd import field as _field
    ops = _field._get_ops()
    trig, stride = env.noisy_field._select_trig(B)
    print("ops.env_step_nograd        %.2f" % t(lambda: ops.env_step_nograd(env.noisy_field, env.sun_pos, act, trig, stride, c)))
    print("ops.env_step_nograd+notify %.2f" % t(lambda: ops.env_step_nograd(env.noisy_field, env.sun_pos, act, trig, stride, c, notify=True)))
    def stepwait():
        r = ops.env_step_nograd(env.noisy_field, env.sun_pos, act, trig, stride, c, notify=True)
        ops.notify_wait(r[-1])
    print("  ... + notify_wait        %.2f" % t(stepwait))
    print("ops.render_nograd          %.2f" % t(lambda: ops.render_nograd(env.noisy_field, env.sun_pos, act, trig, stride, True)))
    e = torch.empty
    print("torch.empty x10            %.2f" % t(lambda: [e(5, device=dev) for _ in range(10)]))
