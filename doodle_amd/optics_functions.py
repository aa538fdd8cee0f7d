"""The reference's four "internals you may reuse" (/root/reference/README.md:203-207), kept importable.

`HelioField.render` here does NOT go through these: its per-ray geometry and the footprint sum are the HIP
kernels behind `include/helio.h` (helio_geometry_fwd / helio_splat_fwd), which never materialise the per-ray
`[M, R, R]` footprints that `gaussian_blur_batch` returns.  These are plain torch tensor functions for scripts
that call the reference's helpers directly — same names, argument meaning, return values and (on the CPU)
the same bits as newenv_rl_test_multi_error.py:46-149; they run on whatever device their arguments live on
and differentiate through torch autograd.  `tests/test_optics_functions.py` chains them the way :356-406
does and checks every stage against the fixtures produced by the reference.
"""
from __future__ import annotations

import torch

_TINY = 1e-9


def _unit_rows(v: torch.Tensor) -> torch.Tensor:
    return v / v.norm(dim=1, keepdim=True).clamp_min(_TINY)


def reflect_vectors(incidents: torch.Tensor, normals: torch.Tensor) -> torch.Tensor:
    """Mirror `incidents` [M,3] (pointing from the mirror to the source) about `normals` [M,3]
    (any length): r = −i + 2 (i·n̂) n̂, evaluated in the reference's order (:46-50)."""
    n_hat = _unit_rows(normals)
    along = -(incidents * n_hat).sum(dim=1, keepdim=True)
    return -incidents - 2 * along * n_hat


def ray_plane_intersection_batch(ray_origins: torch.Tensor, ray_dirs: torch.Tensor, plane_point: torch.Tensor,
                                 plane_normal: torch.Tensor, epsilon=1e-9):
    """Where rays [M,3]+t·[M,3] meet one plane (:52-75).  Returns `(points [M,3], valid [M,1] as 0./1.)`;
    a ray parallel to the plane (|d·n̂| ≤ epsilon) gets the point (0,0,0) and valid 0."""
    n_hat = plane_normal / plane_normal.norm().clamp_min(_TINY)
    closing = (ray_dirs * n_hat).sum(dim=1, keepdim=True)
    hits = closing.abs() > epsilon
    divisor = torch.where(hits, closing, torch.zeros_like(closing) + epsilon)
    t = ((plane_point - ray_origins) * n_hat).sum(dim=1, keepdim=True) / divisor
    t = torch.where(hits, t, torch.zeros_like(t))
    points = ray_origins + t * ray_dirs
    return torch.where(hits, points, torch.zeros_like(points)), hits.float()


def rotate_normals_batch(normals: torch.Tensor, error_angles_mrad: torch.Tensor) -> torch.Tensor:
    """Orientation errors (:78-104): rotate `normals` [M,3] about Up (z) by `error_angles_mrad[:,1]`, then
    about East (x) by `error_angles_mrad[:,0]` (milliradians).  Not renormalised, like the reference."""
    tilt_east, tilt_up = (error_angles_mrad[:, k] * 1e-3 for k in (0, 1))
    ce, se, cu, su = tilt_east.cos(), tilt_east.sin(), tilt_up.cos(), tilt_up.sin()
    nx, ny, nz = normals.unbind(dim=1)
    x1 = cu * nx - su * ny
    y1 = su * nx + cu * ny
    return torch.stack([x1, ce * y1 - se * nz, se * y1 + ce * nz], dim=1)


def gaussian_blur_batch(intersections: torch.Tensor, heliostat_positions: torch.Tensor, plane_origin: torch.Tensor,
                        plane_u: torch.Tensor, plane_v: torch.Tensor, width: float, height: float, resolution: int,
                        sigma_scale: float, valid_mask: torch.Tensor) -> torch.Tensor:
    """One isotropic Gaussian footprint per ray on the receiver grid (:107-149): [M, resolution, resolution],
    dim 1 along `plane_u`, σ = sigma_scale · |intersection − heliostat|.  A ray with `valid_mask` 0 contributes
    exp(0) = 1 to every pixel, as the reference's masking of the differences does.

    Memory is M·resolution²·(3+1) floats — the reason the renderer proper sums footprints in a kernel instead."""
    M, R = intersections.shape[0], resolution
    dev = intersections.device
    reach = (intersections - heliostat_positions).norm(dim=1)
    sigma = (sigma_scale * reach).clamp_min(_TINY).view(M, 1, 1)
    gu, gv = torch.meshgrid(torch.linspace(-width / 2, width / 2, R, device=dev),
                            torch.linspace(-height / 2, height / 2, R, device=dev), indexing="ij")
    pixels = (plane_origin.view(1, 1, 1, 3) + gu.view(1, R, R, 1) * plane_u.view(1, 1, 1, 3)
              + gv.view(1, R, R, 1) * plane_v.view(1, 1, 1, 3))
    offsets = (pixels - intersections.view(M, 1, 1, 3)) * valid_mask.unsqueeze(1).unsqueeze(1)
    return torch.exp(-offsets.pow(2).sum(dim=3) / (2 * sigma.pow(2)).clamp_min(1e-12))
