#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Runs only in the build container, where /root/reference is mounted.  It imports
the reference's torch-only optics module (newenv_rl_test_multi_error.py) and,
through an in-process stand-in for the absent `gymnasium` package, its env
module (test_environment.py), feeds them seeded inputs and records inputs and
outputs as .npz files.  Fixtures are DATA only; no reference source travels.

    python tests/golden/make_golden.py            # regenerate everything

Every fixture stores:  the scene (helios, target, area, normal, sigma_scale,
resolution), the call inputs (sun, action, the two error tensors), the CPU
torch by-products the kernels take as inputs (xs/ys linspace, cos/sin of the
error angles), the reference outputs (image, actual, refl), stage-by-stage
intermediates obtained by calling the reference's free functions
(intersections, mask), and the autograd gradient of a fixed linear loss.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("HELIO_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch

import newenv_rl_test_multi_error as ref  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))
RADIUS = float(np.hypot(1e4, 1e4))
torch.set_num_threads(8)


def _suns(B, gen):
    d = torch.randn(B, 3, generator=gen)
    d = d / d.norm(dim=1, keepdim=True)
    d[:, 2] = d[:, 2].abs()
    return (d * RADIUS).float()


def _scene(N, gen, base=80.0, span=10.0):
    h = torch.rand(N, 3, generator=gen) * span + base
    h[:, 2] = 0
    return h


def _field(helios, normal, sigma_scale, err, R, max_batch, seed,
           target=(0.0, -5.0, 0.0), area=(15.0, 15.0)):
    torch.manual_seed(seed)
    return ref.HelioField(
        heliostat_positions=helios,
        target_position=torch.tensor(target),
        target_area=area,
        target_normal=torch.tensor(normal),
        error_scale_mrad=err,
        sigma_scale=sigma_scale,
        initial_action_noise=0.01,
        resolution=R,
        device="cpu",
        max_batch_size=max_batch,
    )


def _stages(field, sun2d, action, errs):
    """Re-run the reference's free functions stage by stage to expose the
    intersection points and the validity mask (render() does not return them)."""
    B = sun2d.shape[0]
    N = field.num_heliostats
    flats = action.reshape(-1, 3)
    a = ref.rotate_normals_batch(flats, errs.reshape(-1, 2))
    z = torch.nn.functional.leaky_relu(a[:, -1])
    a = a.clone()
    a[:, -1] = z
    a = a / a.norm(dim=1, keepdim=True).clamp_min(1e-9)
    hel = field.heliostat_positions.view(1, N, 3).expand(B, -1, -1)
    inc = (sun2d.view(B, 1, 3) - hel).reshape(-1, 3)
    inc = inc / inc.norm(dim=-1).unsqueeze(1).clamp_min(1e-9)
    r = ref.reflect_vectors(inc, a)
    r = r / r.norm(dim=-1).unsqueeze(1).clamp_min(1e-9)
    inter, mask = ref.ray_plane_intersection_batch(
        hel.reshape(-1, 3), r, field.target_position, field.target_normal)
    return inter, mask


def record(name, field, sun, action, with_grad=True, seed=1234, extra=None):
    """Render through the reference and save one fixture."""
    sun = torch.as_tensor(sun, dtype=torch.float32)
    sun2d = sun if sun.dim() > 1 else sun.unsqueeze(0)
    B = sun2d.shape[0]
    N = field.num_heliostats
    R = field.resolution
    act = torch.as_tensor(action, dtype=torch.float32).clone().requires_grad_(with_grad)

    ideal = field.calculate_ideal_normals(sun)
    img, actual, refl = field.render(sun, act, ideal, monitor=True)

    if B == 1:
        errs = field.error_angles_mrad.unsqueeze(0)
    else:
        errs = field.batch_error_angles_mrad[:B]
    ang = errs * 1e-3
    trig = torch.stack([ang[..., 0].cos(), ang[..., 0].sin(),
                        ang[..., 1].cos(), ang[..., 1].sin()], dim=-1)
    with torch.no_grad():
        inter, mask = _stages(field, sun2d, act.detach().reshape(B, N, 3), errs)

    d = dict(
        helios=field.heliostat_positions.numpy(),
        target_position=field.target_position.numpy(),
        target_normal=field.target_normal.numpy(),
        target_area=np.array([field.target_width, field.target_height], np.float64),
        plane_u=field.plane_u.numpy(), plane_v=field.plane_v.numpy(),
        sigma_scale=np.float64(field.sigma_scale),
        error_scale_mrad=np.float64(field.error_scale_mrad),
        resolution=np.int64(R), max_batch_size=np.int64(field.max_batch_size),
        sun=sun.numpy(), action=act.detach().numpy(),
        error_angles_mrad=field.error_angles_mrad.numpy(),
        batch_error_angles_mrad=(field.batch_error_angles_mrad.numpy()
                                 if field.batch_error_angles_mrad is not None
                                 else np.zeros((0, N, 2), np.float32)),
        trig=trig.numpy(),
        xs=torch.linspace(-field.target_width / 2, field.target_width / 2, R).numpy(),
        ys=torch.linspace(-field.target_height / 2, field.target_height / 2, R).numpy(),
        ideal=ideal.detach().numpy(),
        image=img.detach().numpy(), actual=actual.detach().numpy(),
        refl=refl.detach().numpy(),
        inter=inter.numpy(), mask=mask.numpy(),
    )
    if with_grad:
        g = torch.Generator().manual_seed(seed)
        G = torch.randn(img.shape, generator=g)
        H = torch.randn(actual.shape, generator=g)
        Q = torch.randn(refl.shape, generator=g)
        # three separate cotangents, then the combined one
        (gi,) = torch.autograd.grad((img * G).sum(), act, retain_graph=True)
        (ga,) = torch.autograd.grad((actual * H).sum(), act, retain_graph=True)
        (gr,) = torch.autograd.grad((refl * Q).sum(), act, retain_graph=True)
        (gall,) = torch.autograd.grad((img * G).sum() + (actual * H).sum()
                                      + (refl * Q).sum(), act)
        d.update(G=G.numpy(), H=H.numpy(), Q=Q.numpy(),
                 grad_from_image=gi.numpy(), grad_from_actual=ga.numpy(),
                 grad_from_refl=gr.numpy(), grad_all=gall.numpy())
    if extra:
        d.update(extra)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name:34s} B={B:3d} N={N:3d} R={R:3d}  peak={float(img.max()):.4g}"
          f"  {os.path.getsize(path)/1024:.0f} KiB")


def noisy_ideal(field, sun, gen, noise=0.01):
    ideal = field.calculate_ideal_normals(sun)
    a = ideal + noise * torch.randn(ideal.shape, generator=gen)
    a = a / a.norm(dim=-1, keepdim=True)
    return a.reshape(a.shape[0], -1) if a.dim() == 3 else a.flatten()


def main():
    up = (0.0, 1.0, 0.0)

    # ---- G1: the README / training-script scene ------------------------------
    gen = torch.Generator().manual_seed(0)
    helios = _scene(50, gen)
    suns = _suns(25, gen)

    f = _field(helios, up, 0.01, 90.0, 128, 25, seed=1)
    record("g1_train_n50_b25_r128", f, suns, noisy_ideal(f, suns, gen))

    f = _field(helios, up, 0.1, 180.0, 64, 25, seed=2)
    record("g1_readme_n50_b25_r64", f, suns, noisy_ideal(f, suns, gen))

    f = _field(helios, up, 0.01, 0.0, 64, 25, seed=3)
    record("g1_noerr_n50_b25_r64", f, suns, noisy_ideal(f, suns, gen))

    # single-sun conventions: 1-D sun, and a [1,3] sun (both use error_angles_mrad)
    f = _field(helios, up, 0.05, 30.0, 64, 25, seed=4)
    a1 = noisy_ideal(f, suns[0], gen)
    record("g1_single_1d_n50_r64", f, suns[0], a1)
    record("g1_single_b1_n50_r64", f, suns[:1], a1.view(1, -1))

    # ---- G5: prefix rule — B=2 must equal the first two rows of B=3 ----------
    f = _field(helios, up, 0.05, 60.0, 32, 25, seed=5)
    a3 = noisy_ideal(f, suns[:3], gen)
    record("g5_prefix_b3_n50_r32", f, suns[:3], a3)
    record("g5_prefix_b2_n50_r32", f, suns[:2], a3[:2])

    # ---- G2: tiny hand-checkable scene ---------------------------------------
    h4 = torch.tensor([[80.0, 85.0, 0.0], [84.0, 81.0, 0.0],
                       [88.0, 88.0, 0.0], [82.0, 90.0, 0.0]])
    sun1 = torch.tensor([[5000.0, 6000.0, 11000.0]])
    f = _field(h4, up, 0.01, 0.0, 16, 4, seed=6)
    ideal = f.calculate_ideal_normals(sun1)
    record("g2_tiny_ideal_n4_r16", f, sun1, ideal.reshape(1, -1))
    for tag, shift in (("east", (3.0, 0.0, 0.0)), ("up", (0.0, 0.0, 3.0))):
        # aim at a point shifted on the target plane: the spot must move along
        # image dim0 (East) or dim1 (Up)
        tgt = f.target_position + torch.tensor(shift)
        inc = sun1.view(1, 1, 3) - h4.view(1, 4, 3)
        out = tgt.view(1, 1, 3) - h4.view(1, 4, 3)
        nrm = inc / inc.norm(dim=2, keepdim=True) + out / out.norm(dim=2, keepdim=True)
        nrm = nrm / nrm.norm(dim=2, keepdim=True)
        record(f"g2_tiny_{tag}_n4_r16", f, sun1, nrm.reshape(1, -1))
    # leaky-ReLU path: normals with negative Z
    # (wide spots so the far-flung reflections still leave a non-zero image)
    f = _field(h4, up, 0.6, 0.0, 16, 4, seed=6)
    neg = ideal.clone()
    neg[..., 2] = -neg[..., 2].abs() * torch.tensor([1.0, 0.5, 2.0, 0.1])
    record("g2_tiny_negz_n4_r16", f, sun1, neg.reshape(1, -1))

    # ---- G3: general target normal (plane_v from the cross product) ----------
    f = _field(helios, (0.3, 0.9, -0.2), 0.02, 40.0, 64, 8, seed=7)
    record("g3_tilted_n50_b5_r64", f, suns[:5], noisy_ideal(f, suns[:5], gen))

    # ---- G4: an invalid (plane-parallel) ray: contributes 1.0 to every pixel -
    hpar = torch.tensor([[0.0, 50.0, 0.0], [3.0, 60.0, 0.0]])
    sunp = torch.tensor([[0.0, 50.0, 1000.0], [10.0, 55.0, 2000.0]])
    f = _field(hpar, up, 0.05, 0.0, 16, 2, seed=8)
    actp = f.calculate_ideal_normals(sunp).clone()
    s = 2.0 ** -0.5
    actp[0, 0] = torch.tensor([s, 0.0, s])      # reflects (0,0,1) into (1,0,0): denom == 0
    record("g4_parallel_n2_b2_r16", f, sunp, actp.reshape(2, -1))

    # ---- larger N, odd sizes (ragged tiles) -----------------------------------
    gen2 = torch.Generator().manual_seed(11)
    hel201 = _scene(201, gen2, span=40.0)
    s7 = _suns(7, gen2)
    f = _field(hel201, up, 0.03, 20.0, 48, 7, seed=9)
    record("g8_ragged_n201_b7_r48", f, s7, noisy_ideal(f, s7, gen2))
    f = _field(hel201[:33], (0.1, 1.0, 0.05), 0.02, 10.0, 100, 3, seed=10)
    record("g8_ragged_n33_b3_r100", f, s7[:3], noisy_ideal(f, s7[:3], gen2))

    # ---- ideal normals / init_actions (rows C, D) -----------------------------
    f = _field(helios, up, 0.01, 1.0, 16, 25, seed=12)
    torch.manual_seed(99)
    f.init_actions(suns)
    init_b = f.initial_action.clone()
    torch.manual_seed(99)
    f.init_actions(suns[3])
    init_1 = f.initial_action.clone()
    np.savez_compressed(
        os.path.join(OUT, "g9_ideal_init.npz"),
        helios=helios.numpy(), target_position=f.target_position.numpy(),
        suns=suns.numpy(),
        ideal_batched=f.calculate_ideal_normals(suns).numpy(),
        ideal_single=f.calculate_ideal_normals(suns[3]).numpy(),
        init_seed=np.int64(99), initial_action_noise=np.float64(0.01),
        init_batched=init_b.numpy(), init_single=init_1.numpy())
    print("g9_ideal_init")

    make_env_golden(helios, suns)


def make_env_golden(helios, suns):
    """G6: HelioEnv.reset()/step() observations, metrics and monitors."""
    # gymnasium is not installed; the env only needs Env, spaces.Box, spaces.Dict.
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class _Env:  # noqa: D401
        def __init__(self, *a, **k):
            pass

    class _Box:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class _Dict(dict):
        def __init__(self, d):
            super().__init__(d)

    gym.Env, spaces.Box, spaces.Dict, gym.spaces = _Env, _Box, _Dict, spaces
    sys.modules["gymnasium"], sys.modules["gymnasium.spaces"] = gym, spaces
    import test_environment as refenv  # noqa: E402

    # (tag, sigma_scale, err mrad, use_error_mask, exponential_risk, single_sun, N, B, R, action noise)
    cases = (("train", 0.01, 90.0, False, False, False, 50, 25, 64, 0.003),
             ("readme", 0.1, 180.0, False, False, False, 50, 25, 64, 0.003),
             ("mask", 0.02, 60.0, True, False, False, 50, 25, 64, 0.003),
             # exponential_risk (:464-488) with actions far enough off for rays to leave the target;
             # single_sun (:297-305): the env builds its own repeated sun, set_sun_pos is not called
             ("exprisk", 0.03, 30.0, False, True, False, 20, 12, 48, 0.03),
             ("single", 0.02, 45.0, True, False, True, 20, 12, 48, 0.01))
    for tag, sig, err, masked, exp_risk, single, N, B, R, noise in cases:
        torch.manual_seed(21)
        hel = helios[:N]
        env = refenv.HelioEnv(
            heliostat_pos=hel, targ_pos=torch.tensor([0.0, -5.0, 0.0]),
            targ_area=(15.0, 15.0), targ_norm=torch.tensor([0.0, 1.0, 0.0]),
            sigma_scale=sig, error_scale_mrad=err, initial_action_noise=0.0,
            resolution=R, batch_size=B, device="cpu",
            new_errors_every_reset=False, use_error_mask=masked, error_mask_ratio=0.2,
            exponential_risk=exp_risk, single_sun=single, azimuth=30.0 if single else 45.0,
            elevation=50.0 if single else 45.0)
        cone_suns = env.sun_pos.clone()
        if not single:
            env.set_sun_pos(suns[:B])
        obs0 = env.reset()
        gen = torch.Generator().manual_seed(5)
        act = env.ideal_normals + noise * torch.randn(env.ideal_normals.shape, generator=gen)
        act = (act / act.norm(dim=2, keepdim=True)).reshape(B, -1).requires_grad_(True)
        obs, metrics, monitor = env.step(act)
        grads = {}
        for k in ("mse", "dist", "bound", "alignment_loss"):
            (g,) = torch.autograd.grad(metrics[k], act, retain_graph=True, allow_unused=True)
            grads["grad_" + k] = (g if g is not None else torch.zeros_like(act)).numpy()
        np.savez_compressed(
            os.path.join(OUT, f"g6_env_{tag}_n{N}_b{B}_r{R}.npz"),
            helios=hel.numpy(), sigma_scale=np.float64(sig),
            error_scale_mrad=np.float64(err), resolution=np.int64(R), batch_size=np.int64(B),
            use_error_mask=np.bool_(masked), exponential_risk=np.bool_(exp_risk), single_sun=np.bool_(single),
            azimuth=np.float64(30.0 if single else 45.0), elevation=np.float64(50.0 if single else 45.0),
            cone_suns=cone_suns.numpy(), suns=env.sun_pos.numpy(),
            error_angles_mrad=env.noisy_field.error_angles_mrad.numpy(),
            batch_error_angles_mrad=env.noisy_field.batch_error_angles_mrad.numpy(),
            distance_maps=env.distance_maps.numpy(),
            ref_min=env.ref_min.numpy(), ref_max=env.ref_max.numpy(),
            reset_img=obs0["img"].numpy(), reset_aux=obs0["aux"].numpy(),
            action=act.detach().numpy(),
            step_img=obs["img"].detach().numpy(), step_aux=obs["aux"].detach().numpy(),
            **{"metric_" + k: v.detach().numpy() for k, v in metrics.items()},
            **{"monitor_" + k: v.detach().numpy() for k, v in monitor.items()},
            **grads)
        print(f"g6_env_{tag}", {k: float(v) for k, v in metrics.items()},
              "rays off target:", int((monitor["all_bounds"] > 0).sum()))


if __name__ == "__main__":
    main()
