"""Parity of the HIP path (through the C ABI) against the golden fixtures made by the
reference.  Needs an MI355X: run with ``-m gpu``.

Bars (BASELINE.json north_star):
  * ``actual`` and ``refl``: BIT-EXACT with the reference's CPU fp32 results;
  * image: ``allclose(rtol=1e-5, atol=1e-8)`` per pixel AND max|Δ| ≤ 1e-5·peak;
  * grad_action: max|Δ| ≤ GRAD_BAR·max|grad| — per backward variant, 2x the worst deviation measured over the
    fixtures (8e-7 … 2.7e-6; profiles/r04_a_grad_floor.txt).  The ACCURACY of the gradient — against the same formulas in
    float64, next to the reference's own fp32 error of 2e-5 — is tests/test_grad_accuracy_gpu.py.
"""
import numpy as np
import pytest
import torch

from conftest import check_grad, golden, render_fixture_names
from grad_floor import RENDER_BAR as GRAD_BAR

pytestmark = pytest.mark.gpu
NAMES = render_fixture_names()
DEV = "cuda"


def field_from(g, variant=None):
    from doodle_amd import HelioField
    f = HelioField(g["helios"], g["target_position"], tuple(float(x) for x in g["target_area"]),
                   g["target_normal"], error_scale_mrad=float(g["error_scale_mrad"]),
                   sigma_scale=float(g["sigma_scale"]), resolution=int(g["resolution"]), device=DEV,
                   max_batch_size=int(g["max_batch_size"]))
    # inject the pre-sampled errors as CPU tensors: their cos/sin then come from torch's
    # CPU kernels, exactly as in the reference run that made the fixture
    f.error_angles_mrad = torch.from_numpy(g["error_angles_mrad"])
    f.batch_error_angles_mrad = torch.from_numpy(g["batch_error_angles_mrad"]) if g["batch_error_angles_mrad"].size else None
    return f


def assert_image_close(img, ref):
    img, ref = np.asarray(img), np.asarray(ref).reshape(np.shape(img))
    np.testing.assert_allclose(img, ref, rtol=1e-5, atol=1e-8)
    assert np.abs(img - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)


def fused_form_exists(variant, N, R):
    """Variants 10..13 force one form of the single-launch render; not every form exists for every size."""
    return {10: N <= 64, 11: N <= 128, 12: N <= 256, 13: N <= 8 and R % 4 == 0}.get(variant, True)


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13])
@pytest.mark.parametrize("name", NAMES)
def test_forward_matches_reference(name, variant, monkeypatch):
    from doodle_amd import native
    g = golden(name)
    if not fused_form_exists(variant, g["helios"].shape[0], int(g["resolution"])):
        pytest.skip("this form of the fused kernel does not exist for the fixture's size")
    f = field_from(g)
    monkeypatch.setattr(native.get_ops(), "splat_variant", variant)
    img, actual, refl = f.render(torch.from_numpy(g["sun"]), torch.from_numpy(g["action"]), None, monitor=True)
    assert tuple(img.shape) == g["image"].shape
    assert tuple(actual.shape) == g["actual"].shape and tuple(refl.shape) == g["refl"].shape
    assert np.array_equal(actual.cpu().numpy(), g["actual"])          # bit-exact
    assert np.array_equal(refl.cpu().numpy(), g["refl"])              # bit-exact
    assert_image_close(img.cpu().numpy(), g["image"])


@pytest.mark.parametrize("name", NAMES)
def test_ray_parameters_match_reference_intersections(name):
    """The (a, b, k2, c2) work buffer restates the reference's intersection points."""
    from doodle_amd import native
    g = golden(name)
    f = field_from(g)
    sun = torch.from_numpy(g["sun"]).reshape(-1, 3).to(DEV)
    B, N = sun.shape[0], f.num_heliostats
    trig = torch.from_numpy(g["trig"]).reshape(B, N, 4).to(DEV).contiguous()
    normals = torch.from_numpy(g["action"]).reshape(B, N, 3).to(DEV).contiguous()
    _, _, rays = native.get_ops().geometry_fwd(f.heliostat_positions, sun, normals, trig, 4 * N, f._plane)
    rays = rays.cpu().numpy().reshape(-1, 4)
    x, mask = g["inter"], g["mask"][:, 0]
    d0 = g["target_position"][None, :] - x
    assert np.array_equal(rays[:, 0], (d0[:, 0] * g["plane_u"][0] + d0[:, 1] * g["plane_u"][1]) + d0[:, 2] * g["plane_u"][2])
    assert np.array_equal(rays[:, 1], (d0[:, 0] * g["plane_v"][0] + d0[:, 1] * g["plane_v"][1]) + d0[:, 2] * g["plane_v"][2])
    assert np.array_equal(rays[:, 2] > 0, mask > 0)


@pytest.mark.parametrize("bwd_variant", [1, 2, 3, 4, 5, 6, 7, 8, 12])
@pytest.mark.parametrize("name", NAMES)
def test_backward_matches_reference_autograd(name, bwd_variant, monkeypatch):
    from doodle_amd import native
    g = golden(name)
    f = field_from(g)
    monkeypatch.setattr(native.get_ops(), "bwd_variant", bwd_variant)
    act = torch.from_numpy(g["action"]).to(DEV).requires_grad_(True)
    img, actual, refl = f.render(torch.from_numpy(g["sun"]), act, None, monitor=True)
    G, H, Q = (torch.from_numpy(g[k]).to(DEV) for k in ("G", "H", "Q"))
    for loss, key in (((img * G.reshape(img.shape)).sum(), "grad_from_image"),
                      ((actual * H).sum(), "grad_from_actual"),
                      ((refl * Q).sum(), "grad_from_refl")):
        (ga,) = torch.autograd.grad(loss, act, retain_graph=True)
        check_grad(ga, g[key], GRAD_BAR[bwd_variant], key)
    (ga,) = torch.autograd.grad((img * G.reshape(img.shape)).sum() + (actual * H).sum() + (refl * Q).sum(), act)
    check_grad(ga, g["grad_all"], GRAD_BAR[bwd_variant], "grad_all")


def test_ideal_normals_bit_exact():
    from doodle_amd import HelioField
    g = golden("g9_ideal_init")
    f = HelioField(g["helios"], g["target_position"], (15.0, 15.0), [0.0, 1.0, 0.0], device=DEV)
    suns = torch.from_numpy(g["suns"])
    assert np.array_equal(f.calculate_ideal_normals(suns).cpu().numpy(), g["ideal_batched"])
    assert np.array_equal(f.calculate_ideal_normals(suns[3]).cpu().numpy(), g["ideal_single"])
    f.initial_action_noise = 0.0
    f.init_actions(suns)
    assert tuple(f.initial_action.shape) == (25, 150)
    f.init_actions(suns[0])
    assert tuple(f.initial_action.shape) == (150,)


@pytest.mark.parametrize("name", ["g1_train_n50_b25_r128", "g1_readme_n50_b25_r64", "g3_tilted_n50_b5_r64",
                                  "g8_ragged_n201_b7_r48"])
def test_accuracy_against_fp64_truth_is_at_the_reference_level(name):
    """Accuracy, not only agreement: measured against the same formulas in fp64, the HIP image is
    as close to the truth as the reference's own fp32 image (it inherits the reference's geometry
    bits, and its footprint arithmetic adds no error of its own)."""
    from oracle import torch_oracle as to
    g = golden(name)
    sc64 = to.Scene.build(g["helios"], g["target_position"], tuple(g["target_area"]), g["target_normal"],
                          int(g["resolution"]), float(g["sigma_scale"]), dtype=torch.float64)
    sun = torch.from_numpy(g["sun"]).reshape(-1, 3)
    errs = to.pick_errors(torch.from_numpy(g["error_angles_mrad"]), torch.from_numpy(g["batch_error_angles_mrad"]),
                          sun.shape[0]).double()
    truth, _ = to.render(sc64, sun, torch.from_numpy(g["action"]), errs)
    peak = truth.max().item()
    ref_err = (torch.from_numpy(g["image"]).double().reshape(truth.shape) - truth).abs().max().item() / peak
    f = field_from(g)
    img, _ = f.render(torch.from_numpy(g["sun"]), torch.from_numpy(g["action"]), None)
    our_err = (img.cpu().double().reshape(truth.shape) - truth).abs().max().item() / peak
    assert our_err <= 1.25 * ref_err + 1e-6, (our_err, ref_err)
