"""The oracles (oracle/) against the golden fixtures produced by the reference.

CPU only.  This is what pins the oracle: torch_oracle must reproduce every
fixture bit for bit (forward and autograd); the C oracle must reproduce the
geometry bit for bit and the image within the stated tolerance.
"""
import os

import numpy as np
import pytest
import torch

from conftest import ENV_FIXTURES, golden, render_fixture_names
from oracle import c_oracle, torch_oracle as to

NAMES = render_fixture_names()


def _scene(g):
    return to.Scene.build(g["helios"], g["target_position"], tuple(g["target_area"]),
                          g["target_normal"], int(g["resolution"]), float(g["sigma_scale"]))


def _errs(g, B):
    e = to.pick_errors(torch.from_numpy(g["error_angles_mrad"]),
                       torch.from_numpy(g["batch_error_angles_mrad"]), B)
    assert e is not None
    return e


@pytest.mark.parametrize("name", NAMES)
def test_torch_oracle_forward_bit_exact(name):
    g = golden(name)
    sc = _scene(g)
    sun = torch.from_numpy(g["sun"]).reshape(-1, 3)
    B = sun.shape[0]
    img, actual, refl = to.render(sc, sun, torch.from_numpy(g["action"]), _errs(g, B), monitor=True)
    ref_img = g["image"] if g["sun"].ndim > 1 else g["image"][None]
    assert torch.equal(sc.plane_u, torch.from_numpy(g["plane_u"]))
    assert torch.equal(sc.plane_v, torch.from_numpy(g["plane_v"]))
    assert np.array_equal(actual.numpy(), g["actual"])
    assert np.array_equal(refl.numpy(), g["refl"])
    assert np.array_equal(img.numpy(), ref_img)


@pytest.mark.parametrize("name", NAMES)
def test_torch_oracle_autograd_bit_exact(name):
    g = golden(name)
    sc = _scene(g)
    sun = torch.from_numpy(g["sun"]).reshape(-1, 3)
    B = sun.shape[0]
    act = torch.from_numpy(g["action"]).clone().requires_grad_(True)
    img, actual, refl = to.render(sc, sun, act, _errs(g, B), monitor=True)
    G = torch.from_numpy(g["G"]).reshape(img.shape)
    H, Q = torch.from_numpy(g["H"]), torch.from_numpy(g["Q"])
    (ga,) = torch.autograd.grad((img * G).sum() + (actual * H).sum() + (refl * Q).sum(), act)
    assert np.array_equal(ga.numpy(), g["grad_all"])


@pytest.mark.parametrize("name", NAMES)
def test_c_oracle_geometry_bit_exact_and_image_close(name):
    g = golden(name)
    sun = g["sun"].reshape(-1, 3)
    B, N = sun.shape[0], g["helios"].shape[0]
    actual, refl, inter, mask = c_oracle.geometry(
        g["helios"], sun, g["action"], g["trig"], g["target_position"], g["target_normal"])
    assert np.array_equal(actual, g["actual"].reshape(B, N, 3))
    assert np.array_equal(refl.reshape(-1, 3), g["refl"])
    assert np.array_equal(inter.reshape(-1, 3), g["inter"])
    assert np.array_equal(mask.reshape(-1, 1), g["mask"])
    img = c_oracle.splat(inter, mask, g["helios"], g["target_position"], g["plane_u"], g["plane_v"],
                         g["xs"], g["ys"], float(g["sigma_scale"]))
    ref = g["image"].reshape(B, *img.shape[1:])
    # tolerance: the north-star bound (1e-5 relative, fp32) with an absolute floor
    np.testing.assert_allclose(img, ref, rtol=1e-5, atol=1e-8)
    assert np.abs(img - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)


def test_oracle_chunked_matches_unchunked():
    g = golden("g8_ragged_n201_b7_r48")
    sc = _scene(g)
    sun = torch.from_numpy(g["sun"])
    errs = _errs(g, sun.shape[0])
    img, actual = to.render_chunked(sc, sun, torch.from_numpy(g["action"]), errs, b_chunk=2)
    assert np.array_equal(img.numpy(), g["image"])          # b-chunking preserves bits
    assert np.array_equal(actual.numpy(), g["actual"])
    img2, _ = to.render_chunked(sc, sun, torch.from_numpy(g["action"]), errs, b_chunk=3, n_chunk=64)
    np.testing.assert_allclose(img2.numpy(), g["image"], rtol=2e-6, atol=1e-8)


def test_ideal_normals_and_error_rule():
    g = golden("g9_ideal_init")
    h, tp, suns = (torch.from_numpy(g[k]) for k in ("helios", "target_position", "suns"))
    assert np.array_equal(to.ideal_normals(h, tp, suns).numpy(), g["ideal_batched"])
    assert np.array_equal(to.ideal_normals(h, tp, suns[3]).numpy(), g["ideal_single"])
    assert np.array_equal(c_oracle.ideal_normals(g["helios"], g["suns"], g["target_position"]),
                          g["ideal_batched"])
    single, batch = torch.zeros(5, 2), torch.ones(4, 5, 2)
    assert to.pick_errors(single, batch, 1).shape == (1, 5, 2)
    assert to.pick_errors(single, batch, 1).sum() == 0          # B==1 → the single-sun tensor
    assert to.pick_errors(single, batch, 3).shape == (3, 5, 2)  # prefix slice
    assert to.pick_errors(single, batch, 5) is None             # B > max_batch → fresh sample
    assert to.pick_errors(single, None, 2) is None


def test_prefix_rule_in_goldens():
    g2, g3 = golden("g5_prefix_b2_n50_r32"), golden("g5_prefix_b3_n50_r32")
    assert np.array_equal(g2["image"], g3["image"][:2])
    assert np.array_equal(g2["actual"], g3["actual"][:2])


def test_tiny_scene_properties():
    """Hand-checkable facts about the reference output."""
    ideal = golden("g2_tiny_ideal_n4_r16")["image"][0]
    R = ideal.shape[-1]
    i, j = np.unravel_index(ideal.argmax(), ideal.shape)
    assert {i, j} <= {R // 2 - 1, R // 2}                       # spot at the centre
    east = golden("g2_tiny_east_n4_r16")["image"][0]
    up = golden("g2_tiny_up_n4_r16")["image"][0]
    ie, je = np.unravel_index(east.argmax(), east.shape)
    iu, ju = np.unravel_index(up.argmax(), up.shape)
    assert ie > i + 1 and abs(je - j) <= 1                      # East ↔ image dim0
    assert ju > j + 1 and abs(iu - i) <= 1                      # Up   ↔ image dim1
    par = golden("g4_parallel_n2_b2_r16")
    assert par["mask"].reshape(2, 2)[0, 0] == 0.0               # the plane-parallel ray
    assert par["image"][0].min() >= 1.0                         # contributes 1.0 everywhere


import pytest as _pytest


@_pytest.mark.parametrize("tag", sorted(ENV_FIXTURES))
def test_step_loss_oracle_matches_reference_env(tag):
    """oracle.step_losses on the reference's recorded step() inputs reproduces its metrics,
    monitors and gradients bit for bit (same ATen ops in the same order)."""
    stem, masked, exp_risk, _, _, _ = ENV_FIXTURES[tag]
    g = golden(stem)
    helios = torch.from_numpy(g["helios"])
    tp, tn = torch.tensor([0.0, -5.0, 0.0]), torch.tensor([0.0, 1.0, 0.0])
    sc = to.Scene.build(helios, tp, (15.0, 15.0), tn, int(g["resolution"]), float(g["sigma_scale"]))
    suns = torch.from_numpy(g["suns"])
    act = torch.from_numpy(g["action"]).clone().requires_grad_(True)
    errs = torch.from_numpy(g["batch_error_angles_mrad"])
    ideal = to.ideal_normals(helios, tp, suns)
    img, actual = to.render(sc, suns, act, errs)
    with torch.no_grad():
        target, _ = to.render(sc, suns, ideal.flatten(1), torch.zeros_like(errs))
    out = to.step_losses(img, target, torch.from_numpy(g["distance_maps"]), ideal, actual, act, helios, tp, tn,
                         (15.0, 15.0), exp_risk, error_mask_ratio=0.2 if masked else None)
    names = ("mse", "dist", "bound", "alignment_loss")
    for k, v in zip(names, out[:4]):
        assert np.array_equal(v.detach().numpy(), g["metric_" + k]), k
        (ga,) = torch.autograd.grad(v, act, retain_graph=True, allow_unused=True)
        got = ga.numpy() if ga is not None else np.zeros_like(g["grad_" + k])
        assert np.array_equal(got, g["grad_" + k]), k
    assert np.array_equal(out[4].detach().numpy().reshape(-1, 1), g["monitor_mae_image"])
    assert np.array_equal(out[5].detach().numpy(), g["monitor_all_bounds"])
    assert np.array_equal(out[6].detach().numpy().reshape(-1), g["monitor_alignment_errors"])


def test_reference_fp32_noise_floor_against_fp64():
    """The size of the reference's own fp32 error (~1.8e-5 of peak at
    sigma_scale=0.01): the yardstick for the 1e-5 image tolerance.  The same formulas in fp64
    are the truth."""
    g = golden("g1_train_n50_b25_r128")
    sc64 = to.Scene.build(g["helios"], g["target_position"], tuple(g["target_area"]), g["target_normal"],
                          int(g["resolution"]), float(g["sigma_scale"]), dtype=torch.float64)
    sun = torch.from_numpy(g["sun"])
    img64, _ = to.render(sc64, sun, torch.from_numpy(g["action"]), torch.from_numpy(g["batch_error_angles_mrad"]).double())
    err = (torch.from_numpy(g["image"]).double() - img64).abs().max().item() / img64.max().item()
    assert 1e-6 < err < 1e-4, err


def test_c_oracle_under_address_and_ub_sanitizers(tmp_path):
    """sanitizers run on the CPU build only (GPU ASan is not available on the pool).
    The C oracle, compiled with -fsanitize=address,undefined, runs ragged and degenerate sizes
    without a report."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        _pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_san")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = subprocess.run(["gcc", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=all", os.path.join(root, "oracle", "helio_oracle.c"),
                            os.path.join(root, "tests", "c", "oracle_san.c"), "-lm", "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        _pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "ORACLE SAN OK" in run.stdout, run.stdout + run.stderr


@pytest.mark.parametrize("name", ["g8_ragged_n33_b3_r100", "g1_readme_n50_b25_r64"])
def test_chunked_gradient_oracle_equals_the_reference_autograd(name):
    """oracle.grad_action_chunked (used on the GPU box at N=2000, R=512, where one autograd graph
    does not fit) against the reference's own gradients in the fixtures."""
    g = golden(name)
    sc = _scene(g)
    sun = torch.from_numpy(g["sun"]).reshape(-1, 3)
    B, N, R = sun.shape[0], g["helios"].shape[0], int(g["resolution"])
    G = torch.from_numpy(g["G"]).reshape(B, R, R)
    H = torch.from_numpy(g["H"]).reshape(B, N, 3)
    act = torch.from_numpy(g["action"])
    for h, key in ((None, "grad_from_image"), (H, None)):
        got = to.grad_action_chunked(sc, sun, act, _errs(g, B), G, h, n_chunk=7).reshape(B, -1)
        ref = g[key] if key else g["grad_from_image"] + g["grad_from_actual"]
        ref = ref.reshape(B, -1)
        tol = 0.0 if key else 1e-6          # (two fixture gradients added after the fact round differently)
        assert np.abs(got.numpy() - ref).max() <= tol * np.abs(ref).max(), key
