"""torch_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU restatement, in plain PyTorch tensor ops, of the reference's heliostat
render (DOODLE ``newenv_rl_test_multi_error.py``).  It materialises the same
``[M,R,R,3]`` temporaries with the same ATen ops in the same order, so on CPU it
is bit-identical with the reference in the forward pass AND through autograd;
it is therefore also the oracle for the backward kernels and the thing
``bench.py`` times as ``cpu_baseline`` (kind "port").

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module; ``doodle_amd/`` never does.

Parity status: PINNED by ``tests/test_oracle_golden.py`` against fixtures made
by running the reference itself (``tests/golden/make_golden.py``): image,
``actual``, ``refl`` and ``grad_action`` all compare with ``torch.equal``.

Line numbers cite the reference file ``newenv_rl_test_multi_error.py``.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

_TINY = 1e-9


@dataclass
class Scene:
    """What HelioField.__init__ (:162-216) stores, as plain fp32 CPU tensors."""
    helios: torch.Tensor          # [N,3]
    target_position: torch.Tensor  # [3]
    target_normal: torch.Tensor    # [3] unit (ctor normalises once, :192)
    plane_u: torch.Tensor          # [3]
    plane_v: torch.Tensor          # [3]
    width: float
    height: float
    resolution: int
    sigma_scale: float

    @staticmethod
    def build(helios, target_position, target_area, target_normal, resolution, sigma_scale,
              dtype=torch.float32):
        """``dtype=torch.float64`` gives the same formulas in double: the "truth" the fp32 results
        (the reference's and the kernels') are both measured against in the accuracy tests."""
        helios = torch.as_tensor(helios, dtype=dtype)
        tp = torch.as_tensor(target_position, dtype=dtype)
        tn = torch.as_tensor(target_normal, dtype=dtype)
        tn = tn / tn.norm().clamp_min(_TINY)                          # :192
        u = torch.tensor([1.0, 0.0, 0.0], dtype=dtype)                 # :206
        if torch.allclose(tn, torch.tensor([0.0, 1.0, 0.0], dtype=dtype)):   # :207-210
            v = torch.tensor([0.0, 0.0, 1.0], dtype=dtype)
        else:                                                          # :212-213
            v = torch.linalg.cross(tn, u)
            v = v / v.norm().clamp_min(_TINY)
        return Scene(helios, tp, tn, u, v, float(target_area[0]), float(target_area[1]),
                     int(resolution), float(sigma_scale))


def _rows_to_unit(m: torch.Tensor) -> torch.Tensor:
    return m / m.norm(dim=1, keepdim=True).clamp_min(_TINY)


def tilt(normals: torch.Tensor, err_mrad: torch.Tensor) -> torch.Tensor:
    """Orientation-error rotation of [M,3] normals by [M,2] mrad angles, :78-104:
    about Up (Z) by column 1, then about East (X) by column 0."""
    east = err_mrad[:, 0] * 1e-3
    up = err_mrad[:, 1] * 1e-3
    return tilt_trig(normals, east.cos(), east.sin(), up.cos(), up.sin())


def tilt_trig(normals, ce, se, cu, su) -> torch.Tensor:
    """The rotation of :93-104 with the four trig arrays of :90-91 given."""
    x, y, z = normals[:, 0], normals[:, 1], normals[:, 2]
    xr = cu * x - su * y
    yr = su * x + cu * y
    yt = ce * yr - se * z
    zt = se * yr + ce * z
    return torch.stack([xr, yt, zt], dim=1)


def ray_geometry(scene: Scene, sun: torch.Tensor, normals: torch.Tensor, errs: torch.Tensor, trig=None):
    """sun [B,3], normals [B,N,3], errs [B,N,2] → actual [B,N,3], refl [M,3],
    inter [M,3], mask [M,1], origins [M,3].  Follows :356-389.  ``trig`` [B,N,4] =
    (cos_e, sin_e, cos_u, sin_u) may replace ``errs`` (the kernels take the table)."""
    B, N = normals.shape[0], scene.helios.shape[0]
    if trig is not None:
        t = trig.reshape(-1, 4)
        tilted = tilt_trig(normals.reshape(-1, 3), t[:, 0], t[:, 1], t[:, 2], t[:, 3])
    else:
        tilted = tilt(normals.reshape(-1, 3), errs.reshape(-1, 2))      # :359
    z = torch.nn.functional.leaky_relu(tilted[:, -1])                   # :369
    tilted = tilted.clone()
    tilted[:, -1] = z                                                   # :371
    actual = _rows_to_unit(tilted).view(B, N, 3)                        # :372-373

    origins = scene.helios.view(1, N, 3).expand(B, -1, -1)              # :376
    inc = (sun.view(B, 1, 3) - origins).reshape(-1, 3)                  # :377-379
    inc = inc / inc.norm(dim=-1).unsqueeze(1).clamp_min(_TINY)          # :380

    nhat = _rows_to_unit(actual.reshape(-1, 3))                         # :48
    d = -(inc * nhat).sum(dim=1, keepdim=True)                          # :49
    out = -inc - 2 * d * nhat                                           # :50
    refl = out / out.norm(dim=-1).unsqueeze(1).clamp_min(_TINY)         # :383

    origins = origins.reshape(-1, 3)
    phat = scene.target_normal / scene.target_normal.norm().clamp_min(_TINY)   # :60
    denom = (refl * phat).sum(dim=1, keepdim=True)                      # :62
    ok = denom.abs() > 1e-9                                             # :63
    denom_safe = torch.where(ok, denom, torch.zeros_like(denom) + 1e-9)  # :65
    t = ((scene.target_position - origins) * phat).sum(dim=1, keepdim=True) / denom_safe  # :67
    t = torch.where(ok, t, torch.zeros_like(t))                         # :69
    hit = origins + t * refl                                            # :71
    hit = torch.where(ok, hit, torch.zeros_like(hit))                   # :73
    return actual, refl, hit, ok.float(), origins


def footprints(scene: Scene, hit: torch.Tensor, origins: torch.Tensor, mask: torch.Tensor):
    """[M,3] hits → [M,R,R] un-normalised Gaussians, :107-149."""
    M, R = hit.shape[0], scene.resolution
    sigma = (scene.sigma_scale * (hit - origins).norm(dim=1)).clamp_min(_TINY).view(M, 1, 1)
    xs = torch.linspace(-scene.width / 2, scene.width / 2, R, dtype=hit.dtype)
    ys = torch.linspace(-scene.height / 2, scene.height / 2, R, dtype=hit.dtype)
    gx, gy = torch.meshgrid(xs, ys, indexing="ij")
    pts = (scene.target_position.view(1, 1, 1, 3)
           + gx.view(1, R, R, 1) * scene.plane_u.view(1, 1, 1, 3)
           + gy.view(1, R, R, 1) * scene.plane_v.view(1, 1, 1, 3))
    diffs = pts - hit.view(M, 1, 1, 3)
    diffs = diffs * mask.unsqueeze(1).unsqueeze(1)
    dist_sq = diffs.pow(2).sum(dim=3)
    return torch.exp(-dist_sq / (2 * sigma.pow(2)).clamp_min(1e-12))


def render(scene: Scene, sun, action, errs, monitor: bool = False):
    """Full render for an explicit error slice ``errs`` [B,N,2]; :326-415.

    Returns ``(images [B,R,R], actual [B,N,3])`` and ``refl [M,3]`` if monitor.
    The caller applies the single-sun ``images[0]`` convention and the
    error-selection rule (:340-353) — see :func:`pick_errors`.
    """
    dt = scene.helios.dtype
    sun = torch.as_tensor(sun, dtype=dt).reshape(-1, 3)
    B, N, R = sun.shape[0], scene.helios.shape[0], scene.resolution
    normals = torch.as_tensor(action, dtype=dt).reshape(B, N, 3)
    actual, refl, hit, mask, origins = ray_geometry(scene, sun, normals, errs.to(dt))
    g = footprints(scene, hit, origins, mask)
    images = g.view(B, N, R, R).sum(dim=1)                               # :404-406
    return (images, actual, refl) if monitor else (images, actual)


def render_chunked(scene: Scene, sun, action, errs, b_chunk: int = 1, n_chunk: int | None = None):
    """Forward render of configurations whose ``[M,R,R,3]`` temporary does not fit
    in memory: loops over sun chunks (bit-preserving) and, if ``n_chunk`` is
    given, over heliostat chunks (changes only the order of the sum over n)."""
    sun = torch.as_tensor(sun, dtype=torch.float32).reshape(-1, 3)
    B, N, R = sun.shape[0], scene.helios.shape[0], scene.resolution
    normals = torch.as_tensor(action, dtype=torch.float32).reshape(B, N, 3)
    images = torch.empty(B, R, R)
    actual = torch.empty(B, N, 3)
    with torch.no_grad():
        for b0 in range(0, B, b_chunk):
            b1 = min(B, b0 + b_chunk)
            a, _, hit, mask, origins = ray_geometry(scene, sun[b0:b1], normals[b0:b1], errs[b0:b1])
            actual[b0:b1] = a
            if n_chunk is None:
                images[b0:b1] = footprints(scene, hit, origins, mask).view(b1 - b0, N, R, R).sum(dim=1)
            else:
                acc = torch.zeros(b1 - b0, R, R)
                hit, mask, origins = (t.view(b1 - b0, N, -1) for t in (hit, mask, origins))
                for n0 in range(0, N, n_chunk):
                    n1 = min(N, n0 + n_chunk)
                    g = footprints(scene, hit[:, n0:n1].reshape(-1, 3),
                                   origins[:, n0:n1].reshape(-1, 3), mask[:, n0:n1].reshape(-1, 1))
                    acc += g.view(b1 - b0, n1 - n0, R, R).sum(dim=1)
                images[b0:b1] = acc
    return images, actual


def grad_action_chunked(scene: Scene, sun, action, errs, G, H=None, n_chunk: int = 25, dtype=torch.float32):
    """d/d(action) of ``(images·G).sum() + (actual·H).sum()`` by the reference's own fp32 autograd
    for configurations whose ``[M,R,R,3]`` graph does not fit in memory.  The loss is additive over
    heliostats (``images = Σ_n gauss_n``, :404-406, and ray n's Gaussian depends on action row n
    only), so the gradient rows of a heliostat chunk come from a render of that chunk alone:
    bit-identical with the unchunked autograd row by row.  ``dtype=torch.float64`` (with a float64
    ``scene``) gives the fp64 truth of the same gradient for the accuracy tests."""
    import dataclasses
    sun = torch.as_tensor(sun, dtype=dtype).reshape(-1, 3)
    B, N = sun.shape[0], scene.helios.shape[0]
    normals = torch.as_tensor(action, dtype=dtype).reshape(B, N, 3)
    errs, G = errs.to(dtype), G.to(dtype)
    H = None if H is None else H.to(dtype)
    grad = torch.empty(B, N, 3, dtype=dtype)
    for b in range(B):
        for n0 in range(0, N, n_chunk):
            n1 = min(N, n0 + n_chunk)
            sub = dataclasses.replace(scene, helios=scene.helios[n0:n1])
            a = normals[b:b + 1, n0:n1].clone().requires_grad_(True)
            img, actual = render(sub, sun[b:b + 1], a, errs[b:b + 1, n0:n1])
            loss = (img * G[b:b + 1]).sum()
            if H is not None:
                loss = loss + (actual * H[b:b + 1, n0:n1]).sum()
            (g,) = torch.autograd.grad(loss, a)
            grad[b, n0:n1] = g[0]
    return grad


def pick_errors(single: torch.Tensor, batch: torch.Tensor | None, B: int):
    """The error-selection rule of render(), :340-353.  Returns None when the
    reference would draw a fresh sample (B > max_batch_size)."""
    if B == 1:
        return single.unsqueeze(0)
    if batch is not None and B <= batch.shape[0]:
        return batch[:B]
    return None


def ideal_normals(helios: torch.Tensor, target_position: torch.Tensor, sun) -> torch.Tensor:
    """calculate_ideal_normals, :256-278 (both the 1-D and the batched branch)."""
    sun = torch.as_tensor(sun, dtype=torch.float32)
    N = helios.shape[0]
    if sun.dim() == 1:
        to_sun = sun.view(1, 3) - helios
        to_tgt = target_position.view(1, 3) - helios
        s = _rows_to_unit(to_sun) + _rows_to_unit(to_tgt)
        return _rows_to_unit(s)
    B = sun.shape[0]
    h = helios.view(1, N, 3)
    to_sun = sun.view(B, 1, 3) - h
    to_tgt = target_position.view(1, 1, 3) - h
    s = (to_sun / to_sun.norm(dim=2, keepdim=True).clamp_min(_TINY)
         + to_tgt / to_tgt.norm(dim=2, keepdim=True).clamp_min(_TINY))
    return s / s.norm(dim=2, keepdim=True).clamp_min(_TINY)


# ------------------------------------------------------------------------------------------
# HelioEnv.step loss block (reference: test_environment.py) — oracle for the fused HIP losses
# ------------------------------------------------------------------------------------------
def angles_mrad(v1: torch.Tensor, v2: torch.Tensor, epsilon: float = 1e-10, clamp_dtype=None) -> torch.Tensor:
    """calculate_angles_mrad, test_environment.py:132-155.  ``clamp_dtype`` (accuracy tests only): the dtype
    the clamp bound ``nextafter(1, 0)`` is formed in — float32 keeps the fp32 path's bound when the rest runs
    in float64; None = the inputs' dtype, as in the reference."""
    cosang = torch.sum(v1 * v2, dim=-1)
    cd = clamp_dtype or cosang.dtype
    one = torch.tensor(1.0, dtype=cd)
    upper = torch.nextafter(one, torch.tensor(0.0, dtype=cd))
    return torch.acos(torch.clamp(cosang, min=(-upper).item() + epsilon, max=upper.item() - epsilon)) * 1000


def boundary_all(vects, heliostat_pos, targ_pos, targ_norm, targ_area):
    """boundary(..., return_all=True), test_environment.py:101-130, with the axes step() passes
    (:461-462): east (1,0,0), up (0,0,1)."""
    u = torch.tensor([1.0, 0.0, 0.0], dtype=vects.dtype)
    v = torch.tensor([0.0, 0.0, 1.0], dtype=vects.dtype)
    tol = 0.75
    dots = torch.einsum('bij,j->bi', -vects, targ_norm)
    valid = dots.abs() > 1e-6
    t = torch.einsum('j,bij->bi', targ_pos, vects) / (dots + (~valid).float() * 1e-6)
    inter = heliostat_pos.unsqueeze(0) + vects * t.unsqueeze(2)
    local = inter - targ_pos
    xl = torch.einsum('bij,j->bi', local, u)
    yl = torch.einsum('bij,j->bi', local, v)
    hw, hh = (targ_area[0] * tol) / 2, (targ_area[1] * tol) / 2
    dx = torch.relu(xl.abs() - hw * tol)
    dy = torch.relu(yl.abs() - hh * tol)
    dist = torch.sqrt(dx * dx + dy * dy + 1e-8)
    inside = (xl.abs() <= hw) & (yl.abs() <= hh) & valid
    return dist * (~inside).float()


def step_losses(img, target, distance_maps, ideal, actual, action, heliostat_pos, targ_pos, targ_norm,
                targ_area, exponential_risk: bool = False, error_mask_ratio=None, clamp_dtype=None):
    """The loss block of HelioEnv.step, test_environment.py:436-488; ``error_mask_ratio`` selects
    the use_error_mask=True branch (:444-452: only the worst images, by torch.quantile of their
    mean error, enter mse and dist).
    Returns (mse, dist, bound, alignment_loss, mae_image [B], all_bounds [B,N], angles [B,N])."""
    tx = target.amax((1, 2), keepdim=True).clamp_min(1e-6)
    pred_n, targ_n = img / tx, target / tx
    err = (pred_n - targ_n).abs()
    mae = err.mean(dim=[-2, -1])
    ang = angles_mrad(ideal, actual, clamp_dtype=clamp_dtype)
    if error_mask_ratio is None:
        mse = torch.nn.functional.mse_loss(pred_n, targ_n)
        dist_l = (err * distance_maps).sum((1, 2)).mean()
    else:
        cutoff = torch.quantile(mae, 1 - error_mask_ratio)
        keep = (mae > cutoff).float().unsqueeze(-1).unsqueeze(-1)
        mse = torch.nn.functional.mse_loss(pred_n * keep, targ_n * keep)
        dist_l = (keep * (err * distance_maps)).sum((1, 2)).mean()
    normals = action.view(img.shape[0], -1, 3)
    allb = boundary_all(normals, heliostat_pos, targ_pos, targ_norm, targ_area)
    bound = torch.mean(torch.exp(allb + 1e-6)) if exponential_risk else allb.mean()
    return mse, dist_l, bound, torch.mean(ang), mae, allb, ang
