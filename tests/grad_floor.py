"""grad_floor.py — TEST INFRASTRUCTURE: the measured floor of the gradient's deviation.

For every gradient the parity tests check, three numbers against the SAME formulas evaluated in float64
(``oracle.torch_oracle`` with ``Scene.build(dtype=float64)``) — "the truth":

  ref_vs_truth   the reference's own fp32 autograd (the committed fixture, or the fp32 oracle that is pinned to it)
  hip_vs_truth   the HIP kernels' gradient, per backward variant
  hip_vs_ref     what the parity tests bound

all as ``max|Δ| / max|truth|``.  ``tests/test_grad_accuracy_gpu.py`` asserts on them; run as a script
(``python tests/grad_floor.py [out.txt]``, on an MI355X) it prints the table committed under ``profiles/``.

Reference lines measured: newenv_rl_test_multi_error.py:142-148, 404-406 (the footprint and its sum, through
autograd), test_environment.py:132-155, 436-488 (the loss block of ``HelioEnv.step``).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from conftest import ENV_FIXTURES, golden, render_fixture_names  # noqa: E402
from oracle import torch_oracle as to  # noqa: E402

DEV = "cuda"
BWD_VARIANTS = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12)
RENDER_KEYS = ("grad_from_image", "grad_from_actual", "grad_from_refl", "grad_all")
ENV_KEYS = ("mse", "dist", "bound", "alignment_loss")

# Bars on hip_vs_ref (max|hip − ref| / max|ref|), each 2 x the worst value this file measured on an MI355X
# (profiles/r04_a_grad_floor.txt):
#   per backward variant over the 15 render fixtures (0 = by size; 9-11 = the small-tile kernel's forms, measured as 3 / 6 / 7)
RENDER_BAR = {0: 8e-7, 1: 2.7e-6, 2: 2.5e-6, 3: 1.2e-6, 4: 8e-7, 5: 1.9e-6, 6: 1.2e-6, 7: 1.2e-6, 8: 1.2e-6,
              9: 1.2e-6, 10: 1.2e-6, 11: 1.2e-6, 12: 2.5e-6}          # (12 = the bits of 2)
#   per metric over the five env fixtures (the same for every backward variant)
ENV_BAR = {"mse": 1.1e-6, "dist": 6e-7, "bound": 8e-6, "alignment_loss": 4e-5}
#   sun 511 of the full-batch launches of configs 4 and 5
CONFIG_BAR = 1.1e-6


def rel(a, b, scale):
    return float(np.abs(np.asarray(a, dtype=np.float64).reshape(np.shape(b)) - np.asarray(b, dtype=np.float64)).max()
                 / max(scale, 1e-300))


# ---------------------------------------------------------------------------------------------------------
# HelioField.render
# ---------------------------------------------------------------------------------------------------------
def render_truth(g):
    """fp64 truth of the four fixture gradients of one render fixture → {key: float64 array [B, 3N]}."""
    sc = to.Scene.build(g["helios"], g["target_position"], tuple(g["target_area"]), g["target_normal"],
                        int(g["resolution"]), float(g["sigma_scale"]), dtype=torch.float64)
    sun = torch.from_numpy(g["sun"]).reshape(-1, 3).double()
    batch = torch.from_numpy(g["batch_error_angles_mrad"]) if g["batch_error_angles_mrad"].size else None
    errs = to.pick_errors(torch.from_numpy(g["error_angles_mrad"]), batch, sun.shape[0]).double()
    a = torch.from_numpy(g["action"]).double().reshape(sun.shape[0], -1).requires_grad_(True)
    img, actual, refl = to.render(sc, sun, a, errs, monitor=True)
    G, H, Q = (torch.from_numpy(g[k]).double() for k in ("G", "H", "Q"))
    losses = {"grad_from_image": (img * G.reshape(img.shape)).sum(), "grad_from_actual": (actual * H).sum(),
              "grad_from_refl": (refl * Q).sum()}
    losses["grad_all"] = losses["grad_from_image"] + losses["grad_from_actual"] + losses["grad_from_refl"]
    out = {}
    for k, loss in losses.items():
        (ga,) = torch.autograd.grad(loss, a, retain_graph=True)
        out[k] = ga.numpy().reshape(g[k].shape)
    return out


def render_hip(g, bwd_variant):
    """The HIP path's four gradients of one render fixture for one backward variant."""
    from doodle_amd import native
    from test_parity_gpu import field_from
    ops = native.get_ops()
    f = field_from(g)
    prev = ops.bwd_variant
    ops.bwd_variant = bwd_variant
    try:
        act = torch.from_numpy(g["action"]).to(DEV).requires_grad_(True)
        img, actual, refl = f.render(torch.from_numpy(g["sun"]), act, None, monitor=True)
        G, H, Q = (torch.from_numpy(g[k]).to(DEV) for k in ("G", "H", "Q"))
        li, la, lr = (img * G.reshape(img.shape)).sum(), (actual * H).sum(), (refl * Q).sum()
        out = {}
        for k, loss in (("grad_from_image", li), ("grad_from_actual", la), ("grad_from_refl", lr),
                        ("grad_all", li + la + lr)):
            (ga,) = torch.autograd.grad(loss, act, retain_graph=True)
            out[k] = ga.cpu().numpy().reshape(g[k].shape)
    finally:
        ops.bwd_variant = prev
    return out


def render_rows(names=None, variants=BWD_VARIANTS):
    """→ rows (fixture, key, variant, ref_vs_truth, hip_vs_truth, hip_vs_ref)."""
    rows = []
    for name in names or render_fixture_names():
        g = golden(name)
        truth = render_truth(g)
        for v in variants:
            if v == 8 and int(g["resolution"]) > 256:
                continue
            hip = render_hip(g, v)
            for k in RENDER_KEYS:
                s = float(np.abs(truth[k]).max())
                rows.append((name, k, v, rel(g[k], truth[k], s), rel(hip[k], truth[k], s), rel(hip[k], g[k], s)))
    return rows


# ---------------------------------------------------------------------------------------------------------
# HelioEnv.step
# ---------------------------------------------------------------------------------------------------------
TP, TN, AREA = (0.0, -5.0, 0.0), (0.0, 1.0, 0.0), (15.0, 15.0)


def env_constants(g):
    """What the reference's step() holds constant while it differentiates (fp32, the reference's bits): ideal
    normals, the error-free field's image of them, the distance maps."""
    N, R = g["helios"].shape[0], int(g["resolution"])
    sc = to.Scene.build(g["helios"], TP, AREA, TN, R, float(g["sigma_scale"]))
    suns = torch.from_numpy(g["suns"])
    ideal = to.ideal_normals(sc.helios, sc.target_position, suns)
    with torch.no_grad():
        target, _ = to.render(sc, suns, ideal.flatten(1), torch.zeros(suns.shape[0], N, 2))
    return sc, suns, ideal, target, torch.from_numpy(g["distance_maps"])


def env_truth(g, masked, exp_risk):
    """fp64 truth of d metric / d action for the four metrics of one env fixture, and the fp64 angles."""
    sc32, suns, ideal, target, dmaps = env_constants(g)
    R = int(g["resolution"])
    sc = to.Scene.build(g["helios"], TP, AREA, TN, R, float(g["sigma_scale"]), dtype=torch.float64)
    B = suns.shape[0]
    errs = to.pick_errors(torch.from_numpy(g["error_angles_mrad"]), torch.from_numpy(g["batch_error_angles_mrad"]), B)
    a = torch.from_numpy(g["action"]).double().requires_grad_(True)
    img, actual = to.render(sc, suns.double(), a, errs.double())
    res = to.step_losses(img, target.double(), dmaps.double(), ideal.double(), actual, a, sc.helios, sc.target_position,
                         sc.target_normal, AREA, exp_risk, 0.2 if masked else None, clamp_dtype=torch.float32)
    out = {}
    for k, m in zip(ENV_KEYS, res[:4]):
        (ga,) = torch.autograd.grad(m, a, retain_graph=True, allow_unused=True)
        out[k] = np.zeros(g["grad_" + k].shape) if ga is None else ga.numpy().reshape(g["grad_" + k].shape)
    return out, res[6].detach().numpy().reshape(-1)


def env_hip(g, tag, bwd_variant=0):
    from doodle_amd import native
    from doodle_amd.env import HelioEnv
    stem, masked, exp_risk, single, az, el = ENV_FIXTURES[tag]
    N, B, R = g["helios"].shape[0], g["suns"].shape[0], int(g["resolution"])
    env = HelioEnv(heliostat_pos=torch.from_numpy(g["helios"]).to(DEV), targ_pos=torch.tensor(TP, device=DEV),
                   targ_area=AREA, targ_norm=torch.tensor(TN, device=DEV), sigma_scale=float(g["sigma_scale"]),
                   error_scale_mrad=float(g["error_scale_mrad"]), initial_action_noise=0.0, resolution=R, batch_size=B,
                   device=DEV, new_errors_every_reset=False, use_error_mask=masked, error_mask_ratio=0.2,
                   exponential_risk=exp_risk, single_sun=single, azimuth=az, elevation=el)
    env.noisy_field.error_angles_mrad = torch.from_numpy(g["error_angles_mrad"])
    env.noisy_field.batch_error_angles_mrad = torch.from_numpy(g["batch_error_angles_mrad"])
    env.set_sun_pos(torch.from_numpy(g["suns"]).to(DEV))
    env.reset()
    env.distance_maps = torch.from_numpy(g["distance_maps"]).to(DEV)
    ops = native.get_ops()
    prev = ops.bwd_variant
    ops.bwd_variant = bwd_variant
    try:
        act = torch.from_numpy(g["action"]).to(DEV).requires_grad_(True)
        _, metrics, monitor = env.step(act)
        out = {}
        for k in ENV_KEYS:
            (ga,) = torch.autograd.grad(metrics[k], act, retain_graph=True, allow_unused=True)
            ref = g["grad_" + k]
            out[k] = np.zeros_like(ref) if ga is None else ga.cpu().numpy().reshape(ref.shape)
    finally:
        ops.bwd_variant = prev
    return out, monitor["alignment_errors"].detach().cpu().numpy().reshape(-1), {k: metrics[k].item() for k in ENV_KEYS}


def env_rows(tags=None, variants=(0, 1, 4)):
    """→ (gradient rows (fixture, metric, variant, ref_vs_truth, hip_vs_truth, hip_vs_ref), angle rows, notes)."""
    rows, angle_rows, notes = [], [], []
    for tag in tags or sorted(ENV_FIXTURES):
        stem, masked, exp_risk = ENV_FIXTURES[tag][:3]
        g = golden(stem)
        truth, ang64 = env_truth(g, masked, exp_risk)
        for v in variants:
            hip, ang_hip, _ = env_hip(g, tag, v)
            for k in ENV_KEYS:
                s = float(np.abs(truth[k]).max())
                if s == 0.0:
                    continue
                rows.append((stem, k, v, rel(g["grad_" + k], truth[k], s), rel(hip[k], truth[k], s),
                             rel(hip[k], g["grad_" + k], s)))
        # the acos-conditioned quantity, ray by ray (test_environment.py:132-155), mrad
        ang_ref = g["monitor_alignment_errors"].astype(np.float64)
        angle_rows.append((stem, float(np.abs(ang_ref - ang64).max()), float(np.abs(ang_hip - ang64).max()),
                           float(np.abs(ang_hip - ang_ref).max()), float(np.abs(ang64).max())))
        notes.append(worst_angle_ray(g, ang_hip, ang64, stem))
    return rows, angle_rows, notes


def worst_angle_ray(g, ang_hip, ang64, stem):
    """The ray where the HIP angle is furthest from the reference's: its clamped cosine c0 (fp32, the reference's
    summation order), both angles, both derivatives d angle / d c, and how many of the fixture's rays have a cosine
    whose sequential fp32 evaluation ((p0+p1)+p2, what the kernel does) differs from torch's ``sum(dim=-1)``."""
    ideal = torch.from_numpy(g["monitor_ideal_normals"]).reshape(-1, 3)
    # `actual` is bit-exact with the reference (tests/test_parity_gpu.py); the fixture holds the action, so redo it
    sc32, suns, _, _, _ = env_constants(g)
    errs = to.pick_errors(torch.from_numpy(g["error_angles_mrad"]), torch.from_numpy(g["batch_error_angles_mrad"]),
                          suns.shape[0])
    actual = to.ray_geometry(sc32, suns, torch.from_numpy(g["action"]).reshape(suns.shape[0], -1, 3), errs)[0].reshape(-1, 3)
    c_torch = torch.sum(ideal * actual, dim=-1)
    p = ideal * actual
    c_seq = (p[:, 0] + p[:, 1]) + p[:, 2]
    differ = int((c_torch != c_seq).sum())
    ang_ref = g["monitor_alignment_errors"].astype(np.float64)
    i = int(np.abs(ang_hip - ang_ref).argmax())
    c0 = float(c_torch[i])
    d_ref = -1000.0 / np.sqrt(max(1.0 - np.float64(np.float32(c0)) ** 2, 1e-300))
    ulp = float(np.spacing(np.float32(ang_ref[i])))
    return (f"{stem}: worst ray {i}: c0 = {c0!r}  angle ref {ang_ref[i]:.7f} hip {ang_hip[i]:.7f} truth(fp64 path) {ang64[i]:.7f} mrad"
            f"  |hip-ref| = {abs(ang_hip[i] - ang_ref[i]):.3e} mrad = {abs(ang_hip[i] - ang_ref[i]) / ulp:.2f} ulp of the angle;"
            f"  d angle/d c = {d_ref:.4e} mrad (1 ulp of c moves the angle by {abs(d_ref) * float(np.spacing(np.float32(c0))):.3e} mrad);"
            f"  rays whose (p0+p1)+p2 differs from torch.sum: {differ} of {c_seq.numel()}")


# ---------------------------------------------------------------------------------------------------------
# one sun of BASELINE configs 4 and 5 (the full-batch launches bench.py times)
# ---------------------------------------------------------------------------------------------------------
def config_rows(cfgs=(("cfg4", 0, 0), ("cfg5", 0, 1024)), sun=511):
    """Full 512-sun launch of config 4 / one rank's shard of config 5, gradient of sun ``sun`` against the fp64
    truth and the fp32 oracle (both chunked over heliostats, newenv_rl_test_multi_error.py:404-406)."""
    from doodle_amd import native
    from test_cull_gpu import _full_batch
    ops = native.get_ops()
    rows = []
    for cfg, seed, b_offset in cfgs:
        w, f, sc, suns, errs, act = _full_batch(cfg, seed, b_offset)
        B, N, R = 512, w.N, w.R
        gen = torch.Generator().manual_seed(5)
        G_last, H_last = torch.randn(1, R, R, generator=gen), torch.randn(1, N, 3, generator=gen)
        G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
        H = torch.randn(B, N, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
        G[sun], H[sun] = G_last[0].to(DEV), H_last[0].to(DEV)
        s = slice(sun, sun + 1)
        grad_ref = to.grad_action_chunked(sc, suns[s], act[s], errs[s], G_last, H_last, n_chunk=25).numpy()
        from doodle_amd import synthetic
        sc64 = to.Scene.build(sc.helios, synthetic.TARGET_POSITION, synthetic.TARGET_AREA, synthetic.TARGET_NORMAL,
                              R, w.sigma_scale, dtype=torch.float64)
        truth = to.grad_action_chunked(sc64, suns[s], act[s], errs[s], G_last, H_last, n_chunk=25,
                                       dtype=torch.float64).numpy()
        scale = float(np.abs(truth).max())
        for cull in (True, False):
            ops.cull = cull
            try:
                a_dev = act.to(DEV).requires_grad_(True)
                img, actual = f.render(suns, a_dev, None)
                (grad,) = torch.autograd.grad((img * G).sum() + (actual * H).sum(), a_dev)
            finally:
                ops.cull = True
            got = grad[s].cpu().numpy().reshape(truth.shape)
            rows.append((f"{cfg} N={N} R={R} B=512 sun {sun}", "grad(image·G + actual·H)", "auto/culled" if cull else "auto/dense",
                         rel(grad_ref, truth, scale), rel(got, truth, scale), rel(got, grad_ref, scale)))
            del img, actual, grad, a_dev
        del G, H, f
        torch.cuda.empty_cache()
    return rows


def fmt(rows, head):
    lines = [head, "-" * len(head)]
    for r in rows:
        lines.append(f"{r[0]:<34} {r[1]:<26} {str(r[2]):<12} {r[3]:>12.3e} {r[4]:>12.3e} {r[5]:>12.3e}")
    return lines


def summary(rows):
    """Per (key, variant): worst ref_vs_truth, hip_vs_truth, hip_vs_ref over the fixtures."""
    agg = {}
    for name, k, v, a, b, c in rows:
        x = agg.setdefault((k, v), [0.0, 0.0, 0.0])
        x[0], x[1], x[2] = max(x[0], a), max(x[1], b), max(x[2], c)
    return [("(worst over fixtures)", k, v, *x) for (k, v), x in sorted(agg.items(), key=lambda t: (t[0][0], str(t[0][1])))]


def main(out=None):
    head = f"{'case':<34} {'gradient of':<26} {'bwd variant':<12} {'ref_vs_truth':>12} {'hip_vs_truth':>12} {'hip_vs_ref':>12}"
    lines = ["# Gradient accuracy: max|Δ| / max|truth|, truth = the reference's formulas in float64 (tests/grad_floor.py)",
             "# ref = the reference's fp32 autograd (fixture / pinned fp32 oracle); hip = the HIP backward", ""]
    r = render_rows()
    lines += fmt(r, head) + [""] + fmt(summary(r), head) + [""]
    e, angles, notes = env_rows()
    lines += fmt(e, head) + [""] + fmt(summary(e), head) + [""]
    lines += ["# per-ray alignment angle (mrad): max|ref-truth|, max|hip-truth|, max|hip-ref|, max angle"]
    lines += [f"{a[0]:<34} {a[1]:>12.3e} {a[2]:>12.3e} {a[3]:>12.3e} {a[4]:>10.3f}" for a in angles] + [""] + notes + [""]
    lines += fmt(config_rows(), head)
    text = "\n".join(lines)
    print(text)
    if out:
        with open(out, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
