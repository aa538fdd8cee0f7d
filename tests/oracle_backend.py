"""A CPU stand-in for doodle_amd.native.HipOps, built on the oracle — TEST ONLY.

It has the same tensor-level interface as HipOps and the same decomposition as the HIP
kernels (ray geometry → (a,b,k2,c2) → separable footprint sum; five moments → ray
cotangents → autograd of the geometry), but every step is plain CPU torch on top of
``oracle.torch_oracle``.  The CPU test-suite uses it to exercise the HOST logic of
HelioField / HelioEnv / the sharded renderer without a GPU, and to check the
decomposition's mathematics against the golden fixtures.  The product never imports it.
"""
import torch

from oracle import torch_oracle as to

LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453


def _scene(helios, plane):
    return to.Scene(helios, torch.tensor(list(plane.origin)), torch.tensor(list(plane.normal)),
                    torch.tensor(list(plane.u)), torch.tensor(list(plane.v)), 0.0, 0.0, 0, float(plane.sigma_scale))


def _rays(sc, plane, hit, mask, origins):
    w = torch.tensor(list(plane.w))
    sigma = (sc.sigma_scale * (hit - origins).norm(dim=1)).clamp_min(1e-9)
    two_s2 = (2 * sigma.pow(2)).clamp_min(1e-12)
    d0 = sc.target_position - hit
    a, b, c = (d0 * sc.plane_u).sum(1), (d0 * sc.plane_v).sum(1), (d0 * w).sum(1)
    k2 = torch.where(mask[:, 0] > 0, LOG2E / two_s2, torch.zeros_like(two_s2))
    return torch.stack([a, b, k2, c * c], dim=1)


class OracleOps:
    arch = "cpu-oracle"
    splat_variant = 0

    def _forward(self, helios, sun, normals, trig, trig_b_stride, plane):
        B, N = normals.shape[0], normals.shape[1]
        sc = _scene(helios, plane)
        tg = trig.reshape(-1, N, 4)
        tg = tg[:B] if trig_b_stride else tg[:1].expand(B, N, 4)
        actual, refl, hit, mask, origins = to.ray_geometry(sc, sun, normals, None, trig=tg)
        return actual, refl.view(B, N, 3), _rays(sc, plane, hit, mask, origins).view(B, N, 4)

    def geometry_fwd(self, helios, sun, normals, trig, trig_b_stride, plane, want_refl=True, want_rays=True):
        with torch.no_grad():
            actual, refl, rays = self._forward(helios, sun, normals, trig, trig_b_stride, plane)
        return actual.contiguous(), (refl.contiguous() if want_refl else None), (rays.contiguous() if want_rays else None)

    def render_fwd(self, helios, sun, normals, trig, trig_b_stride, plane, xs, ys, want_refl=True, rays=None,
                   variant=None):
        actual, refl, r = self.geometry_fwd(helios, sun, normals, trig, trig_b_stride, plane, want_refl, True)
        return self.splat_fwd(r, xs, ys), actual, refl, r

    @staticmethod
    def _factors(rays, xs, ys):
        a, b, k2, c2 = (rays[..., i:i + 1] for i in range(4))
        t, s = xs.view(1, 1, -1) + a, ys.view(1, 1, -1) + b
        return t, s, torch.exp2(-(t * t + c2) * k2), torch.exp2(-(s * s) * k2)

    def splat_fwd(self, rays, xs, ys, variant=None):
        # one image at a time: the per-image arithmetic (SIMD lanes of exp2, matmul
        # blocking) must not depend on the batch — the sharding tests compare sharded and
        # unsharded renders bit for bit, as the HIP kernels guarantee
        out = []
        for b in range(rays.shape[0]):
            _, _, A, E = self._factors(rays[b:b + 1].clone(), xs, ys)
            out.append(A[0].t().contiguous() @ E[0])
        return torch.stack(out).contiguous()

    def splat_bwd(self, rays, xs, ys, grad_image, variant=None):
        t, s, A, E = self._factors(rays.double(), xs.double(), ys.double())
        G = grad_image.double()
        U = lambda wa, we: torch.einsum("bni,bij,bnj->bn", wa, G, we)  # noqa: E731
        m = torch.stack([U(A, E), U(t * A, E), U(A, s * E), U(t * t * A, E), U(A, s * s * E)], dim=-1)
        return m.float().unsqueeze(1).contiguous()       # one column block

    def render_bwd(self, helios, sun, normals, trig, trig_b_stride, plane, rays, xs, ys, grad_image, grad_actual,
                   grad_refl):
        moments = self.splat_bwd(rays, xs, ys, grad_image) if grad_image is not None else None
        return self.geometry_bwd(helios, sun, normals, trig, trig_b_stride, plane, moments, grad_actual, grad_refl)

    def geometry_bwd(self, helios, sun, normals, trig, trig_b_stride, plane, moments, grad_actual, grad_refl):
        with torch.enable_grad():     # called from inside an autograd backward
            n = normals.detach().clone().requires_grad_(True)
            actual, refl, rays = self._forward(helios, sun, n, trig, trig_b_stride, plane)
        outs, cots = [], []
        if moments is not None:
            M = moments.sum(dim=1)
            k2, c2 = rays[..., 2].detach(), rays[..., 3].detach()
            g = torch.stack([-2 * LN2 * k2 * M[..., 1], -2 * LN2 * k2 * M[..., 2],
                             -LN2 * (M[..., 3] + M[..., 4] + c2 * M[..., 0]), -LN2 * k2 * M[..., 0]], dim=-1)
            outs.append(rays)
            cots.append(g)
        if grad_actual is not None:
            outs.append(actual)
            cots.append(grad_actual)
        if grad_refl is not None:
            outs.append(refl)
            cots.append(grad_refl.reshape(refl.shape))
        if not outs:
            return torch.zeros_like(normals)
        (gn,) = torch.autograd.grad(outs, n, cots)
        return gn

    # -- HelioEnv.step loss block: the oracle's restatement plus torch autograd ---------------
    @staticmethod
    def _losses(img, actual, action, c):
        return to.step_losses(img, c.target, c.dmaps, c.ideal, actual, action, c.helios,
                              torch.tensor(list(c.tp)), torch.tensor(list(c.tn)), (c.W, c.H), c.exp_risk,
                              c.mask_ratio if c.mask_ratio >= 0 else None)

    def step_losses_fwd(self, img, actual, action, c):
        with torch.no_grad():
            mse, dist, bound, align, mae, allb, ang = self._losses(img, actual, action, c)
        bad = ~torch.isfinite(torch.stack([mse, dist, bound])).all()
        return torch.stack([mse, dist, bound, align, bad.float()]), mae, ang, allb, torch.ones_like(mae)

    def env_step_fwd(self, helios, sun, normals, trig, trig_b_stride, plane, xs, ys, c, rays=None, want_aux=False,
                     notify=False):
        image, actual, refl, r = self.render_fwd(helios, sun, normals, trig, trig_b_stride, plane, xs, ys)
        out, mae, ang, allb, keep = self.step_losses_fwd(image, actual, normals, c)
        aux = torch.cat([sun, normals.reshape(sun.shape[0], -1)], dim=1) if want_aux else None
        return image, actual, refl, r, out, mae, ang, allb, keep, aux, 0

    def step_losses_bwd(self, img, actual, action, c, g_mse, g_dist, g_bound, g_align, keep, want_img, want_actual,
                        want_action):
        with torch.enable_grad():
            i, a, n = (t.detach().clone().requires_grad_(True) for t in (img, actual, action))
            outs = self._losses(i, a, n, c)[:4]
            total = sum(o * g for o, g in zip(outs, (g_mse, g_dist, g_bound, g_align)) if g is not None)
            gi, ga, gn = torch.autograd.grad(total, (i, a, n), allow_unused=True)
        z = lambda g, like: torch.zeros_like(like) if g is None else g  # noqa: E731
        return (z(gi, img) if want_img else None, z(ga, actual) if want_actual else None,
                z(gn, action) if want_action else None)

    def distance_maps(self, imgs, thr=0.5):
        """The reference's own implementation (test_environment.py:92-97): scipy on the host."""
        import numpy as np
        from scipy.ndimage import distance_transform_edt
        maps = [distance_transform_edt(1 - (im > thr * im.max()).astype(np.uint8)) for im in imgs.cpu().numpy()]
        return torch.tensor(np.stack(maps), dtype=torch.float32)

    def error_trig(self, errs):
        a = errs.detach().float() * 1e-3
        return torch.stack([a[..., 0].cos(), a[..., 0].sin(), a[..., 1].cos(), a[..., 1].sin()], dim=-1)

    def ideal_normals(self, helios, sun, target_xyz):
        return to.ideal_normals(helios, torch.tensor(list(target_xyz)), sun)

    def init_actions(self, ideal, noise, scale):
        """The reference's own ops (newenv_rl_test_multi_error.py:296-303)."""
        noisy = ideal + noise * scale
        return noisy / noisy.norm(dim=-1, keepdim=True).clamp_min(1e-9)


def install(monkeypatch):
    """Route doodle_amd.field through the oracle-backed ops for this test."""
    import doodle_amd.field as field
    ops = OracleOps()
    monkeypatch.setattr(field, "_get_ops", lambda: ops)
    return ops
