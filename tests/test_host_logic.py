"""Host-side logic of the drop-in surface, on CPU, with the oracle-backed ops standing in
for the HIP library (tests/oracle_backend.py).  Also: the C-ABI library loads and exports
every symbol include/helio.h declares, and the product refuses to run without a GPU.
"""
import inspect
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ENV_FIXTURES, ROOT, check_grad, golden, render_fixture_names
import oracle_backend

NAMES = render_fixture_names()
# gradients of the CPU model of the kernels' decomposition (tests/oracle_backend.py) against the reference fixtures:
# max|Δ| / max|ref|, 2x the worst measured in this container (HELIO_RECORD_DEVS, see conftest.check_grad)
CPU_MODEL_BAR = 1.6e-6
CPU_MODEL_ENV_BAR = 9e-7


def field_from(g, device="cpu"):
    from doodle_amd import HelioField
    f = HelioField(g["helios"], g["target_position"], tuple(float(x) for x in g["target_area"]),
                   g["target_normal"], error_scale_mrad=float(g["error_scale_mrad"]),
                   sigma_scale=float(g["sigma_scale"]), resolution=int(g["resolution"]), device=device,
                   max_batch_size=int(g["max_batch_size"]))
    f.error_angles_mrad = torch.from_numpy(g["error_angles_mrad"])
    f.batch_error_angles_mrad = torch.from_numpy(g["batch_error_angles_mrad"]) if g["batch_error_angles_mrad"].size else None
    return f


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from doodle_amd import native
    header = open(os.path.join(ROOT, "include", "helio.h")).read()
    declared = set(re.findall(r"\b(helio_[a-z_0-9]+)\s*\(", header))
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)
    lib = native.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.helio_abi_version() == native.ABI_VERSION
    assert lib.helio_splat_bwd_blocks(128) == 2 and lib.helio_splat_bwd_blocks(100) == 2


def test_comm_library_exports_every_declared_symbol():
    from doodle_amd import comm
    header = open(os.path.join(ROOT, "include", "helio_comm.h")).read()
    declared = set(re.findall(r"\b(helio_(?:comm|p2p)_[a-z_0-9]+)\s*\(", header))
    assert declared == set(comm.COMM_EXPORTS), declared ^ set(comm.COMM_EXPORTS)
    lib = comm.load_comm_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.helio_comm_allgather_f32(None, None, None, 0, None) == -1      # validates before touching RCCL


def test_abi_rejects_bad_arguments_without_launching():
    from doodle_amd import native
    lib = native.load_library()
    plane = native.Plane()
    assert lib.helio_geometry_fwd(0, 5, None, None, None, None, 0, plane, None, None, None, None) == -1
    assert b"bad sizes" in lib.helio_last_error_string()
    assert lib.helio_splat_fwd(1, 1, 8, None, None, None, None, 0, None, 0, None) == -1
    assert lib.helio_render_fwd_launches(25, 50, 128) == 1 and lib.helio_render_fwd_launches(512, 2000, 512) == 2
    assert b"null pointer" in lib.helio_last_error_string()
    # the env-step entry points: size queries and argument checks are host code
    assert lib.helio_env_step_launches(25, 50, 128) == 2 and lib.helio_env_step_launches(512, 2000, 512) == 4
    assert lib.helio_env_step_workspace(25, 50, 128) >= 3 * 25 * 4 + 2 * 25
    assert lib.helio_env_step_workspace(25, 50, 100) >= max(3 * 25 * 4 + 2 * 25, lib.helio_step_losses_workspace(25, 50, 100))
    assert lib.helio_env_step_bwd_image_ws(500, 1, 128) == 0 and lib.helio_env_step_bwd_image_ws(25, 50, 128) == 1
    f3 = (ctypes.c_float * 3)()
    assert lib.helio_env_step_fwd(25, 50, 128, *([None] * 4), 0, plane, *([None] * 6), 0, *([None] * 4), f3, f3, 15.0, 15.0,
                                  0, -1.0, *([None] * 7), None, 0, None, 0, None) == -1
    assert lib.helio_env_step_bwd(25, 50, 128, *([None] * 4), 0, plane, *([None] * 8), f3, f3, 15.0, 15.0, 0,
                                  *([None] * 10), 0, None, 0, None) == -1
    # the optional device scratch (helio.h "Device scratch"): sized by host code; nothing for the small problems
    assert lib.helio_fwd_scratch_bytes(25, 50, 128, 0) == 0 and lib.helio_bwd_scratch_bytes(25, 50, 128, 0) == 0
    t = 4 * 512 * 4                       # int counts[B · tiles²] and int order[…], tiles² = 4 at R = 512 with 256² tiles
    assert lib.helio_fwd_scratch_bytes(512, 2000, 512, 0) == 2 * t + 16 * 512 * 4 * 2000
    assert lib.helio_fwd_scratch_bytes(512, 2000, 512, 6) == 0 and lib.helio_fwd_scratch_bytes(512, 2000, 512, 7) == 0
    lists = 2 * 512 * 2                   # one per (pass, image, 256-wide c tile) at R = 512
    assert lib.helio_bwd_scratch_bytes(512, 2000, 512, 0) == 4 * lists + 4 * lists * 2000 + 256 + 8 * lists * 8 + 8 * lists
    assert lib.helio_bwd_scratch_bytes(512, 5000, 256, 0) == 4 * 512 + 4 * 512 * 5000 + 256 + 8 * 512 * 20 + 8 * 512     # one per image
    # the small-tile kernel: lists only with footprint work enough to carry the launches in front of it
    assert lib.helio_bwd_scratch_bytes(32, 5000, 64, 0) == 0 and lib.helio_bwd_scratch_bytes(256, 5000, 64, 0) > 0
    # the LDS-tile kernels (both passes one launch): lists where that launch is more than one round of the chip
    assert lib.helio_bwd_scratch_bytes(4, 5000, 256, 0) == 0 and lib.helio_bwd_scratch_bytes(16, 5000, 512, 0) > 0
    # … the 64-ray tiles (B = 4, N = 5000, R = 512 by the rules): lists per (pass, c tile), their map in tiles of 64 rays
    assert lib.helio_render_bwd_choice(4, 5000, 512) == 12
    assert lib.helio_bwd_scratch_bytes(4, 5000, 512, 0) == 256 + 4 * 16 * 5000 + 256 + 8 * 16 * 79 + 8 * 16
    assert lib.helio_bwd_scratch_bytes(512, 2000, 512, 5) == 0 and lib.helio_bwd_scratch_bytes(512, 200, 512, 0) == 0
    assert lib.helio_notify_wait(None, 1, 0.0) == -1 and lib.helio_notify_destroy(None) == 0


def test_no_cpu_fallback():
    from doodle_amd import HelioField
    f = HelioField(torch.rand(4, 3), [0.0, -5.0, 0.0], (15.0, 15.0), [0.0, 1.0, 0.0], device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        f.render(torch.rand(3) * 100, torch.rand(12), None)
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        f.calculate_ideal_normals(torch.rand(3) * 100)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "doodle_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn
                assert "libhelio_oracle" not in src, fn


# ------------------------------------------------------------------ surface
def test_constructor_and_method_signatures_match_reference():
    from doodle_amd import HelioField
    from doodle_amd.env import HelioEnv
    p = inspect.signature(HelioField.__init__).parameters
    assert list(p) == ["self", "heliostat_positions", "target_position", "target_area", "target_normal",
                       "error_scale_mrad", "sigma_scale", "initial_action_noise", "resolution", "device",
                       "max_batch_size"]
    assert [p[k].default for k in list(p)[5:]] == [1.0, 0.01, 0.01, 100, "cpu", 25]
    r = inspect.signature(HelioField.render).parameters
    assert list(r) == ["self", "sun_position", "action", "ideal_normals", "show_spillage", "monitor"]
    e = inspect.signature(HelioEnv.__init__).parameters
    assert list(e) == ["self", "heliostat_pos", "targ_pos", "targ_area", "targ_norm", "sigma_scale",
                       "error_scale_mrad", "initial_action_noise", "resolution", "batch_size", "device",
                       "new_sun_pos_every_reset", "new_errors_every_reset", "use_error_mask",
                       "error_mask_ratio", "exponential_risk", "single_sun", "azimuth", "elevation"]
    assert [e[k].default for k in list(e)[5:]] == [0.1, 180.0, 0.0, 128, 25, "cuda", False, True, False, 0.2,
                                                   False, False, 45.0, 45.0]
    f = HelioField(torch.rand(4, 3), [0.0, -5.0, 0.0], (15.0, 12.0), [0.3, 0.9, -0.2], device="cpu")
    for attr in ("device", "max_batch_size", "heliostat_positions", "num_heliostats", "target_position",
                 "target_width", "target_height", "target_normal", "error_scale_mrad", "initial_action_noise",
                 "sigma_scale", "resolution", "error_angles_mrad", "batch_error_angles_mrad", "plane_u",
                 "plane_v", "initial_action"):
        assert hasattr(f, attr), attr
    assert f.error_angles_mrad.shape == (4, 2) and f.batch_error_angles_mrad.shape == (25, 4, 2)
    assert abs(float(f.target_normal.norm()) - 1) < 1e-6 and abs(float((f.plane_u * f.plane_v).sum())) < 1e-7
    assert HelioField(torch.rand(4, 3), [0, -5, 0], (1, 1), [0, 1, 0], max_batch_size=0).batch_error_angles_mrad is None


def test_rng_call_order_matches_reference():
    """Same seed → same error tensors as the reference's constructor (randn(N,2) then
    randn(max_batch,N,2)), checked against a fixture the reference produced."""
    from doodle_amd import HelioField
    g = golden("g1_train_n50_b25_r128")
    torch.manual_seed(1)                                    # seed used by make_golden.py for this field
    f = HelioField(g["helios"], g["target_position"], (15.0, 15.0), g["target_normal"], error_scale_mrad=90.0,
                   sigma_scale=0.01, initial_action_noise=0.01, resolution=128, device="cpu", max_batch_size=25)
    assert np.array_equal(f.error_angles_mrad.numpy(), g["error_angles_mrad"])
    assert np.array_equal(f.batch_error_angles_mrad.numpy(), g["batch_error_angles_mrad"])
    assert np.array_equal(f._xs.numpy(), g["xs"]) and np.array_equal(f._ys.numpy(), g["ys"])


# ------------------------------------------------------------------ render through the CPU model
@pytest.mark.parametrize("name", NAMES)
def test_decomposition_matches_reference(name, monkeypatch):
    """geometry → (a,b,k2,c2) → separable sum, and moments → cotangents → geometry adjoint,
    evaluated on CPU, reproduce the reference's outputs and gradients."""
    oracle_backend.install(monkeypatch)
    g = golden(name)
    f = field_from(g)
    act = torch.from_numpy(g["action"]).clone().requires_grad_(True)
    img, actual, refl = f.render(torch.from_numpy(g["sun"]), act, None, monitor=True)
    assert tuple(img.shape) == g["image"].shape
    assert np.array_equal(actual.detach().numpy(), g["actual"])
    assert np.array_equal(refl.detach().numpy(), g["refl"])
    np.testing.assert_allclose(img.detach().numpy(), g["image"], rtol=1e-5, atol=1e-8)
    G, H, Q = (torch.from_numpy(g[k]) for k in ("G", "H", "Q"))
    for loss, key in (((img * G.reshape(img.shape)).sum(), "grad_from_image"),
                      ((actual * H).sum(), "grad_from_actual"), ((refl * Q).sum(), "grad_from_refl")):
        (ga,) = torch.autograd.grad(loss, act, retain_graph=True)
        check_grad(ga, g[key], CPU_MODEL_BAR, key)


@pytest.mark.parametrize("name", ["g1_readme_n50_b25_r64", "g1_single_1d_n50_r64", "g8_ragged_n33_b3_r100"])
def test_render_value_and_grad_matches_reference_gradients(name, monkeypatch):
    """render_value_and_grad (forward + gradient for given cotangents, no autograd graph) against the
    reference's own gradients, and against render + torch.autograd.grad of this package."""
    oracle_backend.install(monkeypatch)
    g = golden(name)
    f = field_from(g)
    sun, act = torch.from_numpy(g["sun"]), torch.from_numpy(g["action"])
    G, H, Q = (torch.from_numpy(g[k]) for k in ("G", "H", "Q"))
    img, actual, grad = f.render_value_and_grad(sun, act, G, H, Q)
    assert tuple(img.shape) == g["image"].shape and np.array_equal(actual.numpy(), g["actual"])
    np.testing.assert_allclose(img.numpy(), g["image"], rtol=1e-5, atol=1e-8)
    ref = g["grad_all"]
    assert grad.shape == (np.atleast_2d(g["sun"]).shape[0], 3 * f.num_heliostats)
    check_grad(grad, ref, CPU_MODEL_BAR, "grad_all")
    a = act.clone().requires_grad_(True)
    i2, a2, r2 = f.render(sun, a, None, monitor=True)
    (g2,) = torch.autograd.grad((i2 * G.reshape(i2.shape)).sum() + (a2 * H).sum() + (r2 * Q).sum(), a)
    assert torch.equal(grad.reshape(g2.shape), g2) and torch.equal(img, i2.detach())
    # any subset of the cotangents
    _, _, g_img_only = f.render_value_and_grad(sun, act, G)
    ref = g["grad_from_image"]
    check_grad(g_img_only, ref, CPU_MODEL_BAR, "grad_from_image")


def test_return_conventions_and_error_selection(monkeypatch):
    oracle_backend.install(monkeypatch)
    g = golden("g1_single_1d_n50_r64")
    f = field_from(g)
    sun1 = torch.from_numpy(g["sun"])
    a1 = torch.from_numpy(g["action"])
    img, actual = f.render(sun1, a1, None)                    # 1-D sun
    assert img.shape == (64, 64) and actual.shape == (1, 50, 3)
    out = f.render(sun1, a1, None, monitor=True)
    assert len(out) == 3 and out[2].shape == (50, 3)
    img_b1, actual_b1 = f.render(sun1[None], a1[None], None)  # [1,3] sun: batched return, single-sun errors
    assert img_b1.shape == (1, 64, 64) and torch.equal(img_b1[0], img)
    # B == 1 uses error_angles_mrad, B >= 2 the prefix of batch_error_angles_mrad
    suns = torch.stack([sun1, sun1 * torch.tensor([1.0, 0.9, 1.1]), sun1 * torch.tensor([0.8, 1.0, 1.0])])
    acts = a1.repeat(3, 1)
    i3, _ = f.render(suns, acts, None)
    i2, _ = f.render(suns[:2], acts[:2], None)
    assert torch.equal(i3[:2], i2)                            # prefix rule
    assert not torch.equal(i3[0], img)                        # batch row 0 ≠ the single-sun tensor
    f.batch_error_angles_mrad = f.error_angles_mrad[None].repeat(3, 1, 1)
    i3b, _ = f.render(suns, acts, None)
    assert torch.equal(i3b[0], img)                           # same errors → same image (cache was invalidated)
    # duplicated suns get different errors → different images
    f.reset_errors()
    dup, _ = f.render(sun1[None].repeat(2, 1), a1[None].repeat(2, 1), None)
    assert not torch.equal(dup[0], dup[1])
    again, _ = f.render(sun1[None].repeat(2, 1), a1[None].repeat(2, 1), None)
    assert torch.equal(dup, again)                            # deterministic until reset_errors()
    f.reset_errors()
    after, _ = f.render(sun1[None].repeat(2, 1), a1[None].repeat(2, 1), None)
    assert not torch.equal(dup, after)
    # B > max_batch_size: fresh errors every call
    f2 = field_from(g)
    f2.max_batch_size = 1
    f2.reset_errors()
    x, _ = f2.render(suns, acts, None)
    y, _ = f2.render(suns, acts, None)
    assert not torch.equal(x, y)
    # in-place edits of the error tensor are seen (cache keyed on the tensor version)
    f3 = field_from(g)
    p, _ = f3.render(sun1, a1, None)
    f3.error_angles_mrad.mul_(0.5)
    q, _ = f3.render(sun1, a1, None)
    assert not torch.equal(p, q)
    # sigma_scale is a live attribute, as in the reference
    wide = f3.sigma_scale * 4
    f3.sigma_scale = wide
    q2, _ = f3.render(sun1, a1, None)
    assert f3.sigma_scale == wide and not torch.equal(q2, q)
    f3.sigma_scale = wide / 4
    # accepts lists / ndarrays / other dtypes like the reference's as_tensor calls
    r, _ = f3.render(sun1.double().numpy(), a1.numpy().tolist(), None)
    assert torch.equal(r, q)


def test_init_actions(monkeypatch):
    oracle_backend.install(monkeypatch)
    from doodle_amd import HelioField
    g = golden("g9_ideal_init")
    f = HelioField(g["helios"], g["target_position"], (15.0, 15.0), [0.0, 1.0, 0.0], error_scale_mrad=1.0,
                   initial_action_noise=float(g["initial_action_noise"]), resolution=16, device="cpu")
    suns = torch.from_numpy(g["suns"])
    assert np.array_equal(f.calculate_ideal_normals(suns).numpy(), g["ideal_batched"])
    assert np.array_equal(f.calculate_ideal_normals(suns[3]).numpy(), g["ideal_single"])
    torch.manual_seed(int(g["init_seed"]))
    f.init_actions(suns)
    assert np.array_equal(f.initial_action.numpy(), g["init_batched"])
    torch.manual_seed(int(g["init_seed"]))
    f.init_actions(suns[3])
    assert np.array_equal(f.initial_action.numpy(), g["init_single"])


# ------------------------------------------------------------------ HelioEnv
@pytest.mark.parametrize("tag", sorted(ENV_FIXTURES))
def test_env_reset_step_match_reference(tag, monkeypatch):
    oracle_backend.install(monkeypatch)
    from doodle_amd.env import HelioEnv
    stem, masked, exp_risk, single, az, el = ENV_FIXTURES[tag]
    g = golden(stem)
    N, B, R = g["helios"].shape[0], g["suns"].shape[0], int(g["resolution"])
    torch.manual_seed(21)
    env = HelioEnv(heliostat_pos=torch.from_numpy(g["helios"]), targ_pos=torch.tensor([0.0, -5.0, 0.0]),
                   targ_area=(15.0, 15.0), targ_norm=torch.tensor([0.0, 1.0, 0.0]),
                   sigma_scale=float(g["sigma_scale"]), error_scale_mrad=float(g["error_scale_mrad"]),
                   initial_action_noise=0.0, resolution=R, batch_size=B, device="cpu",
                   new_errors_every_reset=False, use_error_mask=masked, error_mask_ratio=0.2,
                   exponential_risk=exp_risk, single_sun=single, azimuth=az, elevation=el)
    # same seed, same RNG call order → same cone suns and the same error tensors
    assert np.array_equal(env.sun_pos.numpy(), g["cone_suns"])
    assert np.array_equal(env.noisy_field.batch_error_angles_mrad.numpy(), g["batch_error_angles_mrad"])
    assert env.observation_space["img"].shape == (B, R, R) and env.action_space.shape == (3 * N,)
    if single:          # the env's own repeated sun (:297-305); set_sun_pos was not called by the reference either
        assert np.array_equal(env.sun_pos.numpy(), g["suns"]) and (env.sun_pos == env.sun_pos[0]).all()
    else:
        env.set_sun_pos(torch.from_numpy(g["suns"]))
    np.testing.assert_allclose(env.distance_maps.numpy(), g["distance_maps"], atol=1e-6)
    np.testing.assert_allclose(float(env.ref_max), float(g["ref_max"]), rtol=1e-5)
    obs0 = env.reset()
    assert set(obs0) == {"img", "aux"}
    np.testing.assert_allclose(obs0["img"].numpy(), g["reset_img"], rtol=1e-5, atol=1e-8)
    assert np.array_equal(obs0["aux"].numpy(), g["reset_aux"])
    act = torch.from_numpy(g["action"]).clone().requires_grad_(True)
    obs, metrics, monitor = env.step(act)
    assert set(metrics) == {"mse", "dist", "bound", "alignment_loss"}
    assert set(monitor) == {"normals", "reflected_rays", "ideal_normals", "all_bounds", "mae_image", "alignment_errors"}
    np.testing.assert_allclose(obs["img"].detach().numpy(), g["step_img"], rtol=1e-5, atol=1e-8)
    assert np.array_equal(obs["aux"].detach().numpy(), g["step_aux"])
    for k in metrics:
        np.testing.assert_allclose(float(metrics[k]), float(g["metric_" + k]), rtol=2e-5, atol=1e-7, err_msg=k)
        (ga,) = torch.autograd.grad(metrics[k], act, retain_graph=True, allow_unused=True)
        ref = g["grad_" + k]
        got = ga.numpy() if ga is not None else np.zeros_like(ref)
        if np.any(ref):
            check_grad(got, ref, CPU_MODEL_ENV_BAR, k)
        else:
            assert not np.any(got), k
    for k in monitor:
        np.testing.assert_allclose(monitor[k].detach().numpy(), g["monitor_" + k], rtol=1e-4, atol=2e-3, err_msg=k)
    # ndarray actions are accepted (:411-412); the dead reference branch is refused loudly
    env.step(g["action"])
    env.new_sun_pos_every_reset = True
    with pytest.raises(NotImplementedError):
        env.reset()


def test_synthetic_inputs_are_shard_invariant():
    """A rank that draws only the rows it owns sees exactly the rows of the global batch
    (bench.py --gpus N relies on it for weak scaling with identical per-sun inputs)."""
    from doodle_amd import synthetic
    w = synthetic.Workload("t", N=7, B=10, R=8)
    helios, suns, errs, noise = synthetic.make_inputs(w, seed=3)
    h2, s2, e2, n2 = synthetic.make_inputs(w, seed=3, b_offset=4, b_count=3)
    assert torch.equal(helios, h2) and torch.equal(suns[4:7], s2)
    assert torch.equal(errs[4:7], e2) and torch.equal(noise[4:7], n2)
    assert (suns[:, 2] >= 0).all() and torch.allclose(suns.norm(dim=1), torch.full((10,), synthetic.SUN_RADIUS))
    assert set(synthetic.CONFIGS) == {"cfg1", "cfg2", "cfg4", "cfg5"}


def test_env_surface_the_training_loop_uses(monkeypatch):
    """What train_with_env.py's rollout() touches (:171-216, :235-275): attributes, a [B,N,3] action
    straight from a policy head, observation shapes, history rolling."""
    oracle_backend.install(monkeypatch)
    from doodle_amd.env import HelioEnv
    torch.manual_seed(0)
    N, B, R, k = 6, 4, 16, 3
    hp = torch.rand(N, 3) * 10 + 80
    hp[:, 2] = 0
    env = HelioEnv(heliostat_pos=hp, targ_pos=torch.tensor([0.0, -5.0, 0.0]), targ_area=(15.0, 15.0),
                   targ_norm=torch.tensor([0.0, 1.0, 0.0]), sigma_scale=0.05, error_scale_mrad=5.0,
                   initial_action_noise=0.0, resolution=R, batch_size=B, device="cpu",
                   new_sun_pos_every_reset=False, new_errors_every_reset=True, use_error_mask=False,
                   error_mask_ratio=0.2, exponential_risk=False)
    env.seed(3)
    for attr in ("batch_size", "resolution", "sun_pos", "ref_min", "ref_max", "distance_maps", "num_heliostats",
                 "ref_field", "noisy_field", "action_space", "observation_space", "heliostat_pos", "targ_pos",
                 "targ_area", "targ_norm", "sigma_scale", "error_scale_mrad", "device"):
        assert hasattr(env, attr), attr
    with torch.no_grad():
        obs = env.reset()
    assert env.ideal_normals.shape == (B, N, 3)
    hist = obs["img"].unsqueeze(1).repeat(1, k, 1, 1)                 # rollout(): hist [B,k,R,R]
    w = torch.zeros(N * 3, N * 3, requires_grad=True)                    # a stand-in policy head
    total = 0.0
    for _ in range(3):
        normals = torch.nn.functional.normalize(env.ideal_normals + (obs["aux"][:, 3:] @ w).view(B, N, 3), dim=2)
        obs, losses, monitor = env.step(normals)                         # [B,N,3] action
        assert obs["img"].shape == (B, R, R) and obs["aux"].shape == (B, 3 + 3 * N)
        assert monitor["normals"].shape == (B, N, 3) and monitor["reflected_rays"].shape == (B * N, 3)
        assert monitor["all_bounds"].shape == (B, N) and monitor["mae_image"].shape == (B, 1)
        hist = torch.roll(hist, -1, dims=1)
        hist[:, -1] = obs["img"].detach()
        total = total + losses["alignment_loss"] + 1e-3 * losses["dist"] + losses["mse"] + 1e-3 * losses["bound"]
    total.backward()
    assert w.grad is not None and torch.isfinite(w.grad).all()
    # set_sun_pos with the suns of another env (:255, :275) and numpy actions (:411-412)
    env.set_sun_pos(env.sun_pos * 1.0)
    env.reset()
    o, l, m = env.step(env.ideal_normals.reshape(B, -1).numpy())
    assert float(l["alignment_loss"]) >= 0.0 and o["img"].shape == (B, R, R)


def test_library_switches_change_the_size_rules_they_name():
    """HELIO_CULL=0 / HELIO_SPLIT=0 are read once by the library (README "Switches"): with them the scratch queries
    return 0 and the size rule never answers a split-sum variant — host code, checked without a device, each in a
    process of its own."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from doodle_amd import native; lib = native.load_library(); "
            "print(lib.helio_fwd_scratch_bytes(512, 2000, 512, 0), lib.helio_bwd_scratch_bytes(512, 2000, 512, 0), "
            "lib.helio_render_fwd_choice(32, 5000, 256), lib.helio_fwd_scratch_required(32, 5000, 256, 0), "
            "lib.helio_fwd_scratch_bytes(2, 5000, 512, 0), lib.helio_bwd_scratch_bytes(16, 5000, 256, 0))" % ROOT)

    def run(**env):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stderr[-2000:]
        return [int(x) for x in out.stdout.split()]

    fwd, bwd, choice, required, ksplit, bwd_mid = run()
    assert fwd > 0 and bwd > 0 and choice == 16 and required == 4 * 32 * 8 * 256 * 256 and ksplit > 0 and bwd_mid > 0
    assert run(HELIO_CULL="0") == [0, 0, 16, required, 0, 0]           # dense everywhere; the split sum keeps its partial images
    fwd2, bwd2, choice2, required2, _, _ = run(HELIO_SPLIT="0")
    assert choice2 == 9 and required2 == 0 and fwd2 == fwd and bwd2 == bwd


def test_scratch_size_queries_are_consistent_over_random_sizes():
    """helio_*_scratch_bytes are host code: over random sizes they are never negative, the required part never exceeds
    the whole, the backward size is 0 or exactly one of the layouts of csrc/cull.h (one list per image; one per
    (pass, 256-wide c tile) where an image is 2..8 tiles wide; either with its work map in tiles of 256 rays or — where the
    rules take the 64-ray tiles — of 64), and forcing a variant that takes no lists gives 0."""
    import random
    from doodle_amd import native
    lib = native.load_library()
    rng = random.Random(7)
    pad = lambda n: (n + 255) // 256 * 256  # noqa: E731

    def bwd_layout(B, N, lists_per_image, sets, tile=256):
        T = B * lists_per_image * sets
        return pad(4 * T) + pad(4 * T * N) + 256 + 8 * T * ((N + tile - 1) // tile) + 8 * T

    seen_lists = seen_ctile = seen_tile64 = 0
    for _ in range(3000):
        B = rng.choice([1, 2, 4, 7, 25, 32, 100, 256, 512, 4096])
        N = rng.choice([1, 50, 96, 200, 257, 300, 1000, 1024, 2000, 5000, 20000])
        R = rng.choice([16, 64, 65, 100, 128, 129, 256, 257, 512, 1000, 2048, 2100])
        fwd, req, bwd = lib.helio_fwd_scratch_bytes(B, N, R, 0), lib.helio_fwd_scratch_required(B, N, R, 0), lib.helio_bwd_scratch_bytes(B, N, R, 0)
        assert 0 <= req <= fwd and bwd >= 0 and fwd % 16 == 0
        if bwd:
            ct = -(-R // 256)
            tile = 64 if lib.helio_render_bwd_choice(B, N, R) == 12 else 256
            per_image, per_ctile = bwd_layout(B, N, 1, 1, tile), bwd_layout(B, N, ct, 2, tile)
            assert bwd == per_image or (R > 128 and 2 <= ct <= 8 and bwd == per_ctile), (B, N, R, bwd)
            seen_lists += 1
            seen_ctile += bwd == per_ctile and ct > 1
            seen_tile64 += tile == 64
            assert N > 256 or (tile == 64 and N > 64)          # (the 64-ray tiles are several per image from 65 rays)
        for v in (1, 4, 5, 6, 7, 8):
            assert lib.helio_bwd_scratch_bytes(B, N, R, v) == 0
        for v in (1, 6, 7, 8, 10, 11, 12, 13):
            assert lib.helio_fwd_scratch_bytes(B, N, R, v) == 0
    assert seen_lists > 100 and seen_ctile > 20 and seen_tile64 > 5


def test_the_size_rules_keep_their_own_conditions_over_random_sizes():
    """helio_render_fwd_choice / helio_render_bwd_choice are host code.  Over random sizes: a split heliostat sum (forward
    14..17) only between 24 and 511 tiles of 256², its parts never shorter than 448 rays and never fewer than round 3's
    rule gave (2 / 4 / 8 from 96 / 48 / 24 tiles); the unsplit 256² kernel (5) only from 192 tiles and 200 rays; the 64-ray
    backward tiles (12) only for fields of more than 32 heliostats on images wider than 64 pixels with a few hundred
    workgroups; the bench's configurations stay where they are measured; invalid sizes give 0."""
    import random
    from doodle_amd import native
    lib = native.load_library()
    rng = random.Random(11)
    seen = {v: 0 for v in (5, 12, 14, 15, 16, 17)}
    for _ in range(4000):
        B = rng.choice([1, 2, 4, 7, 16, 25, 32, 48, 64, 100, 128, 256, 384, 500, 512, 1024, 4096])
        N = rng.choice([1, 8, 33, 50, 96, 200, 257, 300, 448, 500, 1000, 1024, 2000, 5000, 20000])
        R = rng.choice([16, 64, 65, 100, 128, 129, 256, 257, 512, 1000, 2048])
        f, b = lib.helio_render_fwd_choice(B, N, R), lib.helio_render_bwd_choice(B, N, R)
        assert f in (3, 5, 6, 9, 10, 11, 12, 13, 14, 15, 16, 17) and b in (2, 4, 8, 9, 10, 11, 12), (B, N, R, f, b)
        t256 = B * (-(-R // 256)) ** 2
        if 14 <= f <= 17:
            S = 2 << (f - 14)
            assert R > 128 and 24 <= t256 < 512 and N // S >= 448 and S >= (2 if t256 >= 96 else 4 if t256 >= 48 else 8), (B, N, R, f)
            assert lib.helio_fwd_scratch_required(B, N, R, 0) > 0
        if f == 5:
            assert R > 128 and N >= 200 and t256 >= 192, (B, N, R)
        if b == 12:
            ct = -(-R // (128 if R <= 128 else 256))
            assert N > 32 and R > 64 and 2 * ct * (-(-N // 64)) * B >= 240, (B, N, R)
        for v in (f, b):
            if v in seen:
                seen[v] += 1
    assert all(n > 10 for n in seen.values()), seen
    assert lib.helio_render_fwd_choice(512, 2000, 512) == 5 and lib.helio_render_fwd_choice(4096, 5000, 256) == 5
    assert lib.helio_render_fwd_choice(512, 5000, 256) == 5 and lib.helio_render_fwd_choice(25, 50, 128) in (10, 11, 12)
    assert lib.helio_render_bwd_choice(512, 2000, 512) == 2 and lib.helio_render_bwd_choice(512, 5000, 256) == 2
    assert lib.helio_render_fwd_choice(0, 50, 128) == 0 and lib.helio_render_bwd_choice(25, 0, 128) == 0 and lib.helio_render_bwd_choice(25, 50, 0) == 0


def test_receiver_attributes_are_live(monkeypatch):
    """The reference reads target_position / target_normal / plane_u / plane_v / target_width / target_height /
    resolution / sigma_scale from the instance at every render (newenv_rl_test_multi_error.py:387-401).  Host logic on
    the CPU model: after an assignment (or an in-place write) the field renders what a field CONSTRUCTED with the new
    value renders, bit for bit; a frame the separable footprint cannot stand for is refused."""
    from doodle_amd import HelioField
    oracle_backend.install(monkeypatch)
    g = golden("g3_tilted_n50_b5_r64")
    sun, act = torch.from_numpy(g["sun"]), torch.from_numpy(g["action"])

    def fresh(**kw):
        args = dict(heliostat_positions=g["helios"], target_position=g["target_position"],
                    target_area=tuple(float(x) for x in g["target_area"]), target_normal=g["target_normal"],
                    error_scale_mrad=float(g["error_scale_mrad"]), sigma_scale=float(g["sigma_scale"]),
                    resolution=int(g["resolution"]), max_batch_size=int(g["max_batch_size"]))
        args.update(kw)
        f = HelioField(**args)
        f.error_angles_mrad = torch.from_numpy(g["error_angles_mrad"])
        f.batch_error_angles_mrad = torch.from_numpy(g["batch_error_angles_mrad"])
        return f

    f = fresh()
    base, _ = f.render(sun, act, None)
    assert np.array_equal(base.numpy(), g["image"]) or np.allclose(base.numpy(), g["image"], rtol=1e-5, atol=1e-8)
    tp = torch.tensor(g["target_position"]) + torch.tensor([0.5, 0.0, -0.25])
    f.target_position = tp.tolist()                                   # lists are accepted, as in the constructor
    assert isinstance(f.target_position, torch.Tensor) and f.target_position.dtype == torch.float32
    moved, _ = f.render(sun, act, None)
    assert torch.equal(moved, fresh(target_position=tp).render(sun, act, None)[0])
    assert not torch.equal(moved, base)
    f.target_position[0] += 1.0                                       # in place
    tp2 = tp + torch.tensor([1.0, 0.0, 0.0])
    assert torch.equal(f.render(sun, act, None)[0], fresh(target_position=tp2).render(sun, act, None)[0])
    f.target_width, f.target_height, f.resolution, f.sigma_scale = 10.0, 8.0, 24, 0.07
    want = fresh(target_position=tp2, target_area=(10.0, 8.0), resolution=24, sigma_scale=0.07).render(sun, act, None)[0]
    got = f.render(sun, act, None)[0]
    assert got.shape == (5, 24, 24) and torch.equal(got, want)
    assert torch.equal(f.calculate_ideal_normals(sun), fresh(target_position=tp2).calculate_ideal_normals(sun))
    u0 = f.plane_u
    f.plane_u = torch.tensor([2.0, 0.0, 0.0])                         # stored, as in the reference; refused when rendered
    with pytest.raises(ValueError, match="orthonormal"):
        f.render(sun, act, None)
    f.plane_u = u0
    with pytest.raises(ValueError, match="shape"):
        f.target_normal = torch.zeros(4)
    # a copy / pickle rebuilds the records from the attributes
    import copy
    assert torch.equal(copy.deepcopy(f).render(sun, act, None)[0], want)
