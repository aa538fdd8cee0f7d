"""Skipping exactly-zero rays (doodle_amd/csrc/cull.h) changes NO bit: the culled kernels against the dense
ones through the C ABI — forward images, backward moments, gradients — on seeded fields at odd sizes, on
hand-made rays that sit on the criterion's threshold, and at BASELINE.json's full config-4 / config-5
launches, where the LAST suns of the batch are also held to the oracle (the launches bench.py times).
The reference evaluates every (ray, pixel) pair: newenv_rl_test_multi_error.py:142-148, :404-406.
Run with ``-m gpu``."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import check_grad
from grad_floor import CONFIG_BAR
from oracle import torch_oracle as to

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bits(t):
    return t.contiguous().view(torch.int32)


def same_bits(a, b):
    return torch.equal(bits(a), bits(b))


def field_and_rays(N, B, R, sigma, err, seed=0, normal=(0.0, 1.0, 0.0), span=10.0):
    """A seeded synthetic field, its suns / actions, and the (a, b, k2, c2) rays the geometry kernel makes."""
    from doodle_amd import HelioField, native, synthetic
    w = synthetic.Workload("t", N=N, B=B, R=R, sigma_scale=sigma, error_scale_mrad=err, span=span)
    helios, suns, errs, noise = synthetic.make_inputs(w, seed)
    f = HelioField(helios, synthetic.TARGET_POSITION, synthetic.TARGET_AREA, normal, error_scale_mrad=err,
                   sigma_scale=sigma, resolution=R, device=DEV, max_batch_size=max(B, 2))
    f.error_angles_mrad = errs[0]
    f.batch_error_angles_mrad = errs if B > 1 else errs.repeat(2, 1, 1)
    ideal = f.calculate_ideal_normals(suns)
    act = ideal + noise.to(DEV)
    act = (act / act.norm(dim=2, keepdim=True)).contiguous()
    ops = native.get_ops()
    trig, stride = f._select_trig(B)
    _, _, rays = ops.geometry_fwd(f.heliostat_positions, suns.to(DEV), act, trig, stride, f._plane)
    return f, suns.to(DEV), act, rays


def bwd_lists_per_image(R, variant):
    """Lists per image and pass (csrc/cull.h): the LDS-tile kernels get one per 256-wide c tile where an image is
    2..8 of them wide, one per image otherwise."""
    ct = -(-R // 256)
    return ct if variant in (2, 12) and R > 128 and 2 <= ct <= 8 else 1


def bwd_scratch_need(B, N, R, variant, per_image=False):
    pad = lambda n: (n + 255) // 256 * 256      # noqa: E731  counts | idx | totals | map | tail map (csrc/cull.h)
    ct = 1 if per_image else bwd_lists_per_image(R, variant)
    T = B * ct * (2 if ct > 1 else 1)
    tile = 64 if variant == 12 else 256          # rays per item of the work map: variant 12 walks its lists in 64-ray tiles
    return pad(4 * T) + pad(4 * T * N) + 256 + 8 * T * ((N + tile - 1) // tile) + 8 * T, T


def splat_bwd_with_list(rays, xs, ys, G, variant, per_image=False):
    """helio_splat_bwd handed enough scratch for its lists whatever the size rule says → (moments, counts, bytes);
    per_image: only enough for one list per image (the LDS-tile kernels then fall back to those)."""
    from doodle_amd import native
    ops = native.get_ops()
    B, N, R = rays.shape[0], rays.shape[1], xs.shape[0]
    need, T = bwd_scratch_need(B, N, R, variant, per_image)
    scratch = torch.full((need,), 0x7F, dtype=torch.uint8, device=rays.device)
    moments = torch.full((B, ops.lib.helio_splat_bwd_blocks(R), N, 5), float("nan"), device=rays.device)
    rc = ops.lib.helio_splat_bwd(B, N, R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), G.data_ptr(), moments.data_ptr(), variant,
                                 scratch.data_ptr(), need, native._stream())
    assert rc == 0, ops.lib.helio_last_error_string()
    return moments, scratch[:4 * T].view(torch.int32).clone(), need


def splat_with_counts(rays, xs, ys, variant):
    """helio_splat_fwd with a scratch buffer of the test's own → (image, live counts per (image, tile) | None)."""
    from doodle_amd import native
    ops = native.get_ops()
    B, N, R = rays.shape[0], rays.shape[1], xs.shape[0]
    n = ops.lib.helio_fwd_scratch_bytes(B, N, R, variant)
    image = torch.empty((B, R, R), dtype=torch.float32, device=rays.device)
    scratch = torch.full((max(n, 1),), 0x7F, dtype=torch.uint8, device=rays.device)
    rc = ops.lib.helio_splat_fwd(B, N, R, rays.data_ptr(), xs.data_ptr(), ys.data_ptr(), image.data_ptr(), variant,
                                 scratch.data_ptr() if n else None, n, native._stream())
    assert rc == 0, ops.lib.helio_last_error_string()
    if not n:
        return image, None
    tiles = {5: 256, 3: 128, 4: 128}[variant]
    t = -(-R // tiles)
    return image, scratch[:4 * B * t * t].view(torch.int32).view(B, t * t).clone()


@pytest.mark.parametrize("N,B,R", [(257, 3, 260), (333, 2, 200), (700, 2, 512), (192, 5, 129), (96, 90, 512)])      # (96 rays: lists only for long calls)
@pytest.mark.parametrize("sigma,err", [(0.01, 90.0), (0.01, 180.0), (0.01, 0.0), (0.1, 90.0), (0.002, 20.0)])
def test_culled_forward_kernels_equal_the_dense_ones_bit_for_bit(N, B, R, sigma, err):
    from doodle_amd import native
    ops = native.get_ops()
    normal = (0.0, 1.0, 0.0) if (N + B) % 2 else (0.2, 0.95, -0.1)
    f, suns, act, rays = field_and_rays(N, B, R, sigma, err, seed=N + R, normal=normal, span=30.0)
    for variant in (3, 4, 5):
        dense = ops.splat_fwd(rays, f._xs, f._ys, variant=variant, cull=False)
        culled, counts = splat_with_counts(rays, f._xs, f._ys, variant)
        assert counts is not None and int(counts.min()) >= 0 and int(counts.max()) <= N
        assert same_bits(culled, dense), (variant, (culled - dense).abs().max().item())
        assert same_bits(ops.splat_fwd(rays, f._xs, f._ys, variant=variant, cull=True), dense)
        if sigma == 0.1:
            assert int(counts.min()) == N                    # every footprint covers the receiver: nothing to skip
        if err == 0.0:
            assert int(counts.sum()) > 0.9 * counts.numel() * N      # aimed at the centre (bar the action noise)
        if sigma == 0.01 and err >= 90.0:
            assert int(counts.sum()) < 0.8 * counts.numel() * N      # and here a good part of the field misses it
    # the kernels that round at chunk or part boundaries, and the split-bf16 ones, take no list
    for variant in (1, 6, 7, 8, 9):
        assert ops.lib.helio_fwd_scratch_bytes(B, N, R, variant) == 0


def threshold_rays(N, B, R, seed):
    """Hand-made (a, b, k2, c2): footprints whose smallest exponent over the image lies within a few units of the
    criterion's threshold (152) on either side — products around 2^-150, denormal and zero pixels —, plus rays on
    the receiver, plane-parallel rays (k2 = 0: 1.0 everywhere), huge and NaN parameters."""
    g = torch.Generator().manual_seed(seed)
    half = 7.5
    k2 = 10.0 ** (torch.rand(B, N, generator=g) * 6 - 3)                    # 1e-3 … 1e3
    want = 130.0 + 40.0 * torch.rand(B, N, generator=g)                     # target exponent of the nearest pixel
    share = torch.rand(B, N, generator=g)                                   # how it splits between the two axes
    dx, dy = torch.sqrt(want * share / k2), torch.sqrt(want * (1 - share) / k2)
    sx = torch.where(torch.rand(B, N, generator=g) < 0.5, -1.0, 1.0)
    sy = torch.where(torch.rand(B, N, generator=g) < 0.5, -1.0, 1.0)
    a, b = sx * (half + dx), sy * (half + dy)                               # xs + a is nearest to 0 at one edge of the image
    c2 = torch.where(torch.rand(B, N, generator=g) < 0.3, torch.rand(B, N, generator=g) * 5.0 / k2, torch.zeros(B, N))
    rays = torch.stack([a, b, k2, c2], dim=2)
    kind = torch.randint(0, 12, (B, N), generator=g)
    rays[kind == 0] = torch.tensor([0.3, -1.2, 0.8, 0.01])                  # a spot on the receiver
    rays[kind == 1] = torch.tensor([1.0e4, -2.0e4, 0.0, 3.0])               # plane-parallel: adds exactly 1.0 per pixel
    rays[kind == 2] = torch.tensor([3.0e18, 1.0, 1.0e6, 0.0])               # q² overflows: factor exactly 0
    nan = float("nan")
    rays[0, 5] = torch.tensor([nan, 0.0, 1.0, 0.0])                         # (image 0 only: a NaN ray makes its image NaN)
    rays[0, 7] = torch.tensor([0.0, 0.0, nan, 0.0])
    # NaN on ONE axis while the other is far off the receiver: the dense kernels' 0·NaN products are NaN, so these
    # must be kept too (tests/c/cull_floor.cpp found a criterion that dropped them)
    rays[0, 9] = torch.tensor([nan, 1.0e9, 1.0, 0.0])
    rays[0, 11] = torch.tensor([1.0e9, nan, 1.0, 0.0])
    rays[0, 13] = torch.tensor([1.0e9, 1.0e9, 1.0, nan])
    return rays.contiguous()


@pytest.mark.parametrize("N,B,R,seed", [(256, 2, 260, 0), (500, 3, 512, 1), (1024, 2, 129, 2)])
def test_rays_on_the_threshold_of_the_criterion(N, B, R, seed):
    from doodle_amd import native
    ops = native.get_ops()
    rays = threshold_rays(N, B, R, seed).to(DEV)
    xs = torch.linspace(-7.5, 7.5, R).to(DEV)
    ys = torch.linspace(-7.5, 7.5, R).to(DEV)
    for variant in (3, 4, 5):
        dense = ops.splat_fwd(rays, xs, ys, variant=variant, cull=False)
        culled, counts = splat_with_counts(rays, xs, ys, variant)
        both_nan = torch.isnan(dense) & torch.isnan(culled)
        assert torch.equal(bits(culled)[~both_nan], bits(dense)[~both_nan]), variant
        assert torch.equal(torch.isnan(dense), torch.isnan(culled))
        assert 0 < int(counts.sum()) < counts.numel() * N              # some are dropped, some are kept
    # … and the backward's criterion (a factor table that is all zero) on the same rays: the same numbers where the
    # dense kernel has numbers, NaN exactly where it has NaN (the moments of the NaN rays)
    G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed)) * 1e6
    dense_m = ops.splat_bwd(rays, xs, ys, G, variant=2, cull=False)
    culled_m, bcounts, _ = splat_bwd_with_list(rays, xs, ys, G, 2)
    assert N <= 256 or 0 < int(bcounts.sum()) < bcounts.numel() * N       # (up to 256 rays: one ray tile per image, no list)
    assert torch.equal(torch.isnan(dense_m), torch.isnan(culled_m)) and torch.isnan(dense_m).any()
    ok = ~torch.isnan(dense_m)
    assert torch.equal(bits(culled_m)[ok], bits(dense_m)[ok])


@pytest.mark.parametrize("N,B,R", [(257, 3, 260), (700, 2, 512), (1000, 2, 128), (300, 4, 200), (520, 2, 600),
                                   (300, 1, 2100)])      # 2100: nine c tiles — past the eight that get lists of their own
@pytest.mark.parametrize("sigma,err", [(0.01, 90.0), (0.01, 180.0), (0.1, 90.0)])
def test_culled_backward_moments_equal_the_dense_ones_bit_for_bit(N, B, R, sigma, err):
    from doodle_amd import native
    ops = native.get_ops()
    f, suns, act, rays = field_and_rays(N, B, R, sigma, err, seed=N + R + 1, span=30.0)
    G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    dense = ops.splat_bwd(rays, f._xs, f._ys, G, variant=2, cull=False)
    # (the size rule hands over a list only where the dense grid is more than one round of the chip; these small
    # cases pass the scratch themselves to exercise the kernels)
    culled, counts, need = splat_bwd_with_list(rays, f._xs, f._ys, G, 2)
    assert ops.lib.helio_bwd_scratch_bytes(B, N, R, 2) in (0, need) and ops.lib.helio_bwd_scratch_bytes(512, 2000, 512, 2) > 0
    assert same_bits(culled, dense), (culled - dense).abs().max().item()
    assert int(counts.min()) >= 0 and int(counts.max()) <= N
    if bwd_lists_per_image(R, 2) > 1:                    # scratch for one list per image only: those are walked, same bits
        fallback, counts1, need1 = splat_bwd_with_list(rays, f._xs, f._ys, G, 2, per_image=True)
        assert need1 < need and counts1.numel() == B and same_bits(fallback, dense)
        assert int(counts.view(2, B, -1).max(dim=2).values.max()) <= int(counts1.max())      # a tile's list ⊆ its image's
    if sigma == 0.01:                                    # some rays ARE dropped: their moments read exactly +0
        assert int(counts.sum()) < counts.numel() * N
        dead = (culled.abs().sum(dim=(1, 3)) == 0)
        assert dead.any()
        assert not (bits(culled)[dead[:, None, :, None].expand_as(culled)] != 0).any()
    else:
        assert int(counts.sum()) == counts.numel() * N
    for variant in (1, 3, 4, 5, 6, 7):
        assert ops.lib.helio_bwd_scratch_bytes(B, N, R, variant) == 0
    # the 64-ray tiles (variant 12) over the same lists, their map in tiles of 64: the same bits again
    culled12, counts12, need12 = splat_bwd_with_list(rays, f._xs, f._ys, G, 12)
    assert ops.lib.helio_bwd_scratch_bytes(B, N, R, 12) in (0, need12) and ops.lib.helio_bwd_scratch_bytes(500, 520, 256, 12) > 0
    assert same_bits(culled12, dense) and torch.equal(counts12, counts)
    if bwd_lists_per_image(R, 12) > 1:
        fallback12, _, _ = splat_bwd_with_list(rays, f._xs, f._ys, G, 12, per_image=True)
        assert same_bits(fallback12, dense)


def test_render_autograd_and_env_step_are_unchanged_by_the_culling():
    """Through the Python surface at a size whose kernels take the lists (128² forward tiles, 256-ray backward
    tiles): render, its autograd gradient and HelioEnv.step with dist.backward(), scratch on and off."""
    from doodle_amd import native
    from doodle_amd.env import HelioEnv
    ops = native.get_ops()
    N, B, R = 600, 96, 256
    f, suns, act, _ = field_and_rays(N, B, R, 0.01, 90.0, seed=11, span=40.0)
    assert ops.lib.helio_fwd_scratch_bytes(B, N, R, 0) > 0 and ops.lib.helio_bwd_scratch_bytes(B, N, R, 0) > 0
    G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    out = {}
    for cull in (True, False):
        ops.cull = cull
        try:
            a = act.reshape(B, -1).clone().requires_grad_(True)
            img, actual, refl = f.render(suns, a, None, monitor=True)
            (grad,) = torch.autograd.grad((img * G).sum() + actual.sum(), a)
            with torch.no_grad():
                img_ng, _ = f.render(suns, a.detach(), None)
            vg = f.render_value_and_grad(suns, a.detach(), grad_image=G, grad_actual=torch.ones(B, N, 3, device=DEV))
            out[cull] = (img.detach(), actual.detach(), refl.detach(), grad, img_ng, vg[0], vg[2])
        finally:
            ops.cull = True
    for x, y in zip(out[True], out[False]):
        assert same_bits(x, y)
    assert same_bits(out[True][0], out[True][4]) and same_bits(out[True][3], out[True][6])

    env = HelioEnv(f.heliostat_positions, torch.tensor([0.0, -5.0, 0.0], device=DEV), (15.0, 15.0),
                   torch.tensor([0.0, 1.0, 0.0], device=DEV), sigma_scale=0.01, error_scale_mrad=90.0, resolution=R,
                   batch_size=B, device=DEV, new_errors_every_reset=False)
    env.set_sun_pos(suns)
    env.reset()
    res = {}
    for cull in (True, False):
        ops.cull = cull
        try:
            a = act.reshape(B, -1).clone().requires_grad_(True)
            obs, metrics, _ = env.step(a)
            (ga,) = torch.autograd.grad(metrics["dist"] + metrics["mse"], a)
            res[cull] = (obs["img"].detach(), metrics["dist"].detach(), metrics["mse"].detach(), ga)
        finally:
            ops.cull = True
    for x, y in zip(res[True], res[False]):
        assert same_bits(x, y)


def _full_batch(cfg, seed, b_offset=0):
    from doodle_amd import HelioField, synthetic
    w = synthetic.CONFIGS[cfg]
    B = 512
    helios, suns, errs, noise = synthetic.make_inputs(w, seed, b_offset=b_offset, b_count=B)
    f = HelioField(helios, synthetic.TARGET_POSITION, synthetic.TARGET_AREA, synthetic.TARGET_NORMAL,
                   error_scale_mrad=w.error_scale_mrad, sigma_scale=w.sigma_scale, resolution=w.R, device=DEV,
                   max_batch_size=B)
    f.batch_error_angles_mrad = errs
    sc = to.Scene.build(helios, synthetic.TARGET_POSITION, synthetic.TARGET_AREA, synthetic.TARGET_NORMAL, w.R, w.sigma_scale)
    ideal = to.ideal_normals(helios, sc.target_position, suns)
    act = ideal + noise
    act = (act / act.norm(dim=2, keepdim=True)).reshape(B, -1)
    return w, f, sc, suns, errs, act


@pytest.mark.parametrize("cfg,seed,b_offset,last", [("cfg4", 0, 0, (509, 510, 511)), ("cfg5", 0, 1024, (511,))])
def test_the_full_batch_launches_bench_times_are_held_to_the_oracle(cfg, seed, b_offset, last):
    """BASELINE config 4 (N=2000, R=512) and one rank's shard of config 5 (N=5000, R=256) at the FULL batch of 512
    suns — the launches bench.py times: 2048 / 512 workgroups of splat_fwd_mfma_tile<4>, image offsets up to
    537 MB, blockIdx.y up to 511.  (a) the culled and the dense launch agree bit for bit on all 512 images;
    (b) rows [0:24] are the bits of a 24-sun launch forced to the same kernel; (c) the LAST suns of the batch
    meet the oracle (rtol 1e-5 / atol 1e-8 and max|Δ| ≤ 1e-5·peak); (d) the full-batch backward
    (splat_bwd_mfma<0/1>, culled and dense bit-identical) gives the oracle's gradient for sun 511
    (max|Δ| ≤ 1.1e-6·max|grad|, 2x the 5.3e-7 measured; oracle chunked over heliostats,
    newenv_rl_test_multi_error.py:404-406; against the float64 truth: tests/test_grad_accuracy_gpu.py)."""
    from doodle_amd import native
    ops = native.get_ops()
    w, f, sc, suns, errs, act = _full_batch(cfg, seed, b_offset)
    B, N, R = 512, w.N, w.R
    assert ops.lib.helio_render_fwd_choice(B, N, R) == 5 and ops.lib.helio_fwd_scratch_bytes(B, N, R, 0) > 0
    a_dev = act.to(DEV).requires_grad_(True)
    img, actual = f.render(suns, a_dev, None)                                     # default: culled tile<4>
    ops.cull = False
    try:
        with torch.no_grad():
            img_dense, _ = f.render(suns, a_dev.detach(), None)
    finally:
        ops.cull = True
    assert same_bits(img.detach(), img_dense)                                     # (a)
    with torch.no_grad():
        img24, _, _ = f.render_rows(suns[:24], a_dev.detach()[:24], 0, B)          # forced to the whole batch's kernel
    assert same_bits(img24, img_dense[:24])                                       # (b)
    del img_dense, img24
    rows = list(last)
    img_o, actual_o = to.render_chunked(sc, suns[rows], act[rows], errs[rows], b_chunk=1, n_chunk=50)
    assert np.array_equal(actual.detach()[rows].cpu().numpy(), actual_o.numpy())
    got = img.detach()[rows].cpu()
    np.testing.assert_allclose(got.numpy(), img_o.numpy(), rtol=1e-5, atol=1e-8)  # (c)
    for k in range(len(rows)):
        assert (got[k] - img_o[k]).abs().max().item() <= 1e-5 * img_o[k].max().item()

    gen = torch.Generator().manual_seed(5)
    G_last, H_last = torch.randn(1, R, R, generator=gen), torch.randn(1, N, 3, generator=gen)
    G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    H = torch.randn(B, N, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    G[511], H[511] = G_last[0].to(DEV), H_last[0].to(DEV)
    loss = (img * G).sum() + (actual * H).sum()
    assert ops.lib.helio_bwd_scratch_bytes(B, N, R, 0) > 0
    (grad,) = torch.autograd.grad(loss, a_dev, retain_graph=True)
    ops.cull = False
    try:
        (grad_dense,) = torch.autograd.grad(loss, a_dev)
    finally:
        ops.cull = True
    assert same_bits(grad, grad_dense)
    assert torch.isfinite(grad).all()
    grad_o = to.grad_action_chunked(sc, suns[511:], act[511:], errs[511:], G_last, H_last, n_chunk=25)
    check_grad(grad[511:], grad_o, CONFIG_BAR, cfg)                                   # (d)


def test_scratch_that_is_missing_or_too_small_runs_the_dense_kernels():
    from doodle_amd import native
    ops = native.get_ops()
    f, suns, act, rays = field_and_rays(400, 3, 260, 0.01, 90.0, seed=4, span=30.0)
    B, N, R = 3, 400, 260
    need = ops.lib.helio_fwd_scratch_bytes(B, N, R, 5)
    dense = ops.splat_fwd(rays, f._xs, f._ys, variant=5, cull=False)
    for nbytes in (0, 256, need - 256, need, need + 4096):
        image = torch.empty_like(dense)
        scratch = torch.full((max(nbytes, 1),), 0xAB, dtype=torch.uint8, device=DEV)
        guard = scratch.clone()
        rc = ops.lib.helio_splat_fwd(B, N, R, rays.data_ptr(), f._xs.data_ptr(), f._ys.data_ptr(), image.data_ptr(), 5,
                                     scratch.data_ptr(), nbytes, native._stream())
        assert rc == 0 and same_bits(image, dense)
        if nbytes < need:
            assert torch.equal(scratch, guard)           # a buffer that is too small is not touched
    misaligned = torch.empty(need + 512, dtype=torch.uint8, device=DEV)
    rc = ops.lib.helio_splat_fwd(B, N, R, rays.data_ptr(), f._xs.data_ptr(), f._ys.data_ptr(), dense.data_ptr(), 5,
                                 ctypes.c_void_p(misaligned.data_ptr() + 16), need, native._stream())
    assert rc == -1 and b"256-byte" in ops.lib.helio_last_error_string()


@pytest.mark.parametrize("N,B,R", [(700, 3, 512), (1000, 5, 260), (5000, 2, 256), (257, 2, 300)])
@pytest.mark.parametrize("sigma,err", [(0.01, 90.0), (0.02, 40.0)])
def test_split_heliostat_sum_variants(N, B, R, sigma, err):
    """Variants 14..17: the 256² LDS-table kernel with the heliostat sum split into 2..16 parts across workgroups,
    partial images in the scratch, added in part order.  (a) culled == dense-with-scratch bit for bit; (b) within
    the summation-order noise of the unsplit kernel's one-level chain (≤ 8e-6·peak at N = 5000) and inside the oracle's tolerance; (c) the bits
    do not depend on the number of images (a shard forced to the variant reproduces the batch's rows); (d) without
    scratch: HELIO_E_SCRATCH, nothing launched."""
    from doodle_amd import native
    from test_gpu_more import make_case
    ops = native.get_ops()
    # what the size rule hands them: tens of images of a large field
    # (round 4: as many parts as give 512 workgroups — the lists of fewer do not balance —, the unsplit kernel from 512 tiles)
    assert [ops.render_choice(b, 5000, 256) for b in (16, 32, 64, 128, 256, 512)] == [9, 16, 16, 15, 14, 5]
    assert ops.render_choice(32, 2000, 512) == 15 and ops.render_choice(32, 2000, 256) == 9 and ops.render_choice(32, 1000, 256) == 9
    f, sc, suns, errs, act = make_case(N, B, R, sigma=sigma, err=err, seed=N + R + 3, span=30.0)
    s_dev = suns.to(DEV)
    normals = act.to(DEV).reshape(B, N, 3).contiguous()
    trig, stride = f._select_trig(B)
    args = (f.heliostat_positions, s_dev, normals, trig, stride, f._plane, f._xs, f._ys)
    with torch.no_grad():
        unsplit = ops.render_fwd(*args, variant=5)[0]
        img_o, _ = to.render_chunked(sc, suns[:1], act[:1], (errs if B > 1 else errs[:1])[:1], b_chunk=1, n_chunk=50)
        peak = unsplit.max().item()
        for variant in (14, 15, 16, 17):
            assert ops.lib.helio_fwd_scratch_required(B, N, R, variant) == ((4 * B * (2 << (variant - 14)) * R * R + 255) // 256) * 256
            culled = ops.render_fwd(*args, variant=variant)[0]
            ops.cull = False
            try:
                dense = ops.render_fwd(*args, variant=variant)[0]                  # scratch for the partial images only
            finally:
                ops.cull = True
            assert same_bits(culled, dense), variant
            assert (culled - unsplit).abs().max().item() <= 8e-6 * peak, variant     # two summation orders of N terms
            np.testing.assert_allclose(culled[:1].cpu().numpy(), img_o.numpy(), rtol=1e-5, atol=1e-8)
            if B > 1:
                part = ops.render_fwd(f.heliostat_positions, s_dev[1:], normals[1:], trig[1:], stride, f._plane, f._xs, f._ys,
                                      variant=variant)[0]
                assert same_bits(part, culled[1:])
    rays = ops.geometry_fwd(f.heliostat_positions, s_dev, normals, trig, stride, f._plane)[2]
    image = torch.full((B, R, R), 7.0, device=DEV)
    rc = ops.lib.helio_splat_fwd(B, N, R, rays.data_ptr(), f._xs.data_ptr(), f._ys.data_ptr(), image.data_ptr(), 15, None, 0,
                                 native._stream())
    torch.cuda.synchronize()
    assert rc == -6 and b"scratch" in ops.lib.helio_last_error_string() and float(image.min()) == 7.0


@pytest.mark.parametrize("N,B,R", [(1500, 6, 512), (5000, 2, 512), (1200, 7, 513), (3000, 12, 256)])
@pytest.mark.parametrize("sigma,err", [(0.01, 90.0), (0.01, 180.0), (0.1, 90.0)])
def test_ksplit_block_kernel_with_lists_equals_the_dense_one_bit_for_bit(N, B, R, sigma, err):
    """One sun over a whole plant (few images, N >= 1024, >= 2^31 (ray, pixel) pairs): the k-split block kernel takes
    one list per (image, part) — its 16 parts are chains from zero over fixed ray ranges, culled inside each range."""
    from doodle_amd import native
    ops = native.get_ops()
    assert ops.render_choice(B, N, R) == 9 and ops.lib.helio_fwd_scratch_bytes(B, N, R, 0) > 0
    assert ops.lib.helio_fwd_scratch_bytes(B, 1000, R, 9) == 0            # below 1024 rays: dense
    assert ops.lib.helio_fwd_scratch_bytes(1, 5000, 256, 9) == 0          # a short kernel: the compaction launch would not pay
    f, suns, act, rays = field_and_rays(N, B, R, sigma, err, seed=N + R + B, span=40.0)
    dense = ops.splat_fwd(rays, f._xs, f._ys, variant=9, cull=False)
    n = ops.lib.helio_fwd_scratch_bytes(B, N, R, 9)
    image = torch.empty_like(dense)
    scratch = torch.full((n,), 0x7F, dtype=torch.uint8, device=DEV)
    rc = ops.lib.helio_splat_fwd(B, N, R, rays.data_ptr(), f._xs.data_ptr(), f._ys.data_ptr(), image.data_ptr(), 9,
                                 scratch.data_ptr(), n, native._stream())
    assert rc == 0 and same_bits(image, dense)
    counts = scratch[:4 * B * 16].view(torch.int32).view(B, 16)
    per = (((N + 15) // 16) + 3) & ~3
    assert int(counts.min()) >= 0 and int(counts.max()) <= per
    if sigma == 0.1:
        assert int(counts.sum()) == B * N
    else:
        assert int(counts.sum()) < 0.8 * B * N
    # the render through the Python surface takes the same path by default
    with torch.no_grad():
        img, _ = f.render(suns, act.reshape(B, -1), None)
    assert same_bits(img if B > 1 else img, dense)


@pytest.mark.parametrize("N,B,R", [(5000, 2, 512), (1500, 6, 512), (3000, 12, 256), (1200, 7, 513)])
@pytest.mark.parametrize("sigma,err", [(0.01, 90.0), (0.01, 180.0), (0.1, 90.0)])
def test_small_tile_backward_with_lists_equals_the_dense_one_bit_for_bit(N, B, R, sigma, err):
    """The backward of few images of many heliostats: the small-tile kernel in its whole-k form walks 256-ray groups
    of the per-image list (where its dense grid is more than one round of the chip); moments bit-identical with the
    dense launch — forced (variant 3) and as the size rule chooses —, the gradient through render_bwd too."""
    from doodle_amd import native
    ops = native.get_ops()
    # the size query (the LDS-tile kernels' two passes are ONE launch): a list where that launch is more than one round
    assert ops.lib.helio_bwd_scratch_bytes(256, 5000, 64, 0) > 0         # R <= 64: the small-tile kernel, 10240 workgroups
    assert ops.lib.helio_bwd_scratch_bytes(1, 5000, 256, 0) == 0 and ops.lib.helio_bwd_scratch_bytes(B, N, R, 6) == 0
    assert ops.lib.helio_bwd_scratch_bytes(2, 5000, 512, 0) == 0         # 80 tiles a pass of the LDS-tile kernel: one round
    assert ops.lib.helio_bwd_scratch_bytes(4, 5000, 512, 0) > 0          # 160 a pass: 320 workgroups, more than one round
    f, suns, act, rays = field_and_rays(N, B, R, sigma, err, seed=N + R + B + 5, span=40.0)
    G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    dense = ops.splat_bwd(rays, f._xs, f._ys, G, variant=3, cull=False)
    culled, counts, _ = splat_bwd_with_list(rays, f._xs, f._ys, G, 3)
    assert same_bits(culled, dense) and (int(counts.sum()) < B * N) == (sigma == 0.01)
    assert same_bits(ops.splat_bwd(rays, f._xs, f._ys, G, variant=3, cull=True), dense)
    assert same_bits(ops.splat_bwd(rays, f._xs, f._ys, G, variant=0, cull=True), ops.splat_bwd(rays, f._xs, f._ys, G, variant=0, cull=False))
    if sigma == 0.01:
        assert (culled.abs().sum(dim=(1, 3)) == 0).any()
    out = {}
    for cull in (True, False):
        ops.cull = cull
        try:
            out[cull] = f.render_value_and_grad(suns, act.reshape(B, -1), grad_image=G)[2]
        finally:
            ops.cull = True
    assert same_bits(out[True], out[False])


def test_fuzz_of_every_list_taking_kernel_against_its_dense_self():
    """tools/fuzz_cull.py: random shapes (odd sizes included), fields (sigma 0.002…0.1, err 0…400 mrad, tilted
    receivers, spans 10…100 m) and cotangent scales; forward variants 3, 4, 5, 9, 14–17 and backward variants 2, 3, 12
    with their lists against themselves without — any differing bit fails.  (300 cases were run once by hand:
    profiles/r03_e_fuzz_cull.txt; 30 here.)"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_cull.py"), "30", "11"], capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0 and "30 cases, 0 differences" in run.stdout, run.stdout[-3000:] + run.stderr[-2000:]


@pytest.mark.parametrize("N,B,R,sigma,err", [(700, 3, 260, 0.01, 90.0), (300, 2, 512, 0.02, 40.0), (257, 2, 132, 0.01, 180.0),
                                             (1000, 5, 288, 0.05, 90.0)])
def test_double_buffered_backward_tile_gives_the_single_buffered_bits(N, B, R, sigma, err, monkeypatch):
    """splat_bwd_mfma_both<…, DB> (round 4: the two LDS tables twice, 32-deep chunks, the next chunk's factors produced in
    the shadow of this chunk's MFMAs, one barrier per chunk) walks the contracted axis in the order of the single-buffered
    body: the moments are the same bits — dense and with the lists, with a ragged last chunk (R % 32 != 0) and rows past the
    image in the last c tile."""
    from doodle_amd import native
    ops = native.get_ops()
    f, suns, act, rays = field_and_rays(N, B, R, sigma, err, seed=N + R, span=30.0)
    G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(R))
    out = {}
    for db in ("1", "0"):
        monkeypatch.setenv("HELIO_BWD_DB", db)
        out[db] = (ops.splat_bwd(rays, f._xs, f._ys, G, variant=2, cull=False), splat_bwd_with_list(rays, f._xs, f._ys, G, 2)[0])
    torch.cuda.synchronize()
    assert same_bits(out["1"][0], out["0"][0]) and same_bits(out["1"][1], out["0"][1])
    assert same_bits(out["1"][0], out["1"][1])                      # (and the lists change no bit of either)
    assert torch.isfinite(out["1"][0]).all() and out["1"][0].abs().max().item() > 0


@pytest.mark.parametrize("N,B,R", [(50, 7, 260), (96, 3, 512), (130, 4, 300), (192, 2, 132), (64, 5, 128), (40, 9, 100),
                                   (33, 2, 129), (300, 3, 128), (577, 2, 256), (1000, 2, 130)])
def test_64_ray_tiles_give_the_256_ray_tiles_bits(N, B, R):
    """Backward variant 12 (round 4: splat_bwd_mfma_both<…, WR = 1>, the LDS-tile kernel in 64-ray tiles — what the rules
    choose for fields of 33–192 heliostats on large or many images) against variant 2: a ray's chain over the contracted
    axis does not know how many rays share its tile — the same bits, with 256- and 128-wide c tiles, 16-byte and dword
    staging (R % 4 != 0), ragged last tiles and chunks."""
    from doodle_amd import native
    ops = native.get_ops()
    f, suns, act, rays = field_and_rays(N, B, R, 0.02, 40.0, seed=N + R, span=20.0)
    G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(N))
    a = ops.splat_bwd(rays, f._xs, f._ys, G, variant=12, cull=False)
    b = ops.splat_bwd(rays, f._xs, f._ys, G, variant=2, cull=False)
    torch.cuda.synchronize()
    assert same_bits(a, b) and torch.isfinite(a).all() and a.abs().max().item() > 0


def test_the_rules_take_the_64_ray_tiles_where_they_were_measured_ahead():
    from doodle_amd import native
    ops = native.get_ops()
    assert [ops.render_bwd_choice(*s) for s in ((256, 50, 512), (500, 128, 512), (256, 128, 256), (500, 50, 128), (60, 192, 256))] == [12] * 5
    # … and not with four tiles, few workgroups, a field of one ray block, or 64-pixel images
    assert 12 not in [ops.render_bwd_choice(*s) for s in ((256, 200, 256), (60, 50, 256), (256, 50, 128), (500, 16, 256), (500, 50, 64))]
    # larger fields: where the 256-ray tiles pad the field or leave the chip partly idle (profiles/r04_m_bwd_tile64_wide.txt) …
    assert [ops.render_bwd_choice(*s) for s in ((500, 300, 128), (128, 300, 256), (4, 5000, 512), (32, 576, 512), (500, 640, 128),
                                                (500, 384, 256))] == [12] * 6
    # … not where they fit (one exact round of workgroups, too few workgroups, long lists — the configs of the bench)
    assert 12 not in [ops.render_bwd_choice(*s) for s in ((32, 448, 512), (32, 300, 128), (256, 1000, 256), (512, 2000, 512), (512, 5000, 256),
                                                           (4096, 5000, 256))]


@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("N", [1, 7, 8, 9, 57, 64, 65, 72, 200, 449])
def test_the_short_last_chunk_of_the_table_kernels_gives_the_full_chunk_s_bits(N, variant):
    """The LDS-table forward kernels (variants 4, 5 and the split forms 14–17, one template) leave out the k-pairs of the
    last 64-ray chunk that are padding alone, 8 rays at a time (round 4).  The same launch with the rays padded BY THE
    CALLER to a whole chunk — with rays whose row factor is exactly zero, as the kernel's own padding — runs the full
    chunk: the images are the same bits."""
    from doodle_amd import native
    ops = native.get_ops()
    B, R = 3, 256
    f, suns, act, rays = field_and_rays(N, B, R, 0.02, 40.0, seed=N, span=20.0)
    padded = torch.zeros(B, (N + 63) // 64 * 64, 4, device=DEV)
    padded[..., 2], padded[..., 3] = 1.0, 1e30
    padded[:, :N] = rays
    a = ops.splat_fwd(rays, f._xs, f._ys, variant=variant, cull=False)
    b = ops.splat_fwd(padded.contiguous(), f._xs, f._ys, variant=variant, cull=False)
    torch.cuda.synchronize()
    assert same_bits(a, b) and torch.isfinite(a).all() and a.max().item() > 0
    # the two table kernels sum an image's rays in the same order, pair by pair, whatever their tile and chunk (256² tiles of
    # 64-ray chunks; 128² tiles of 32-ray chunks since round 4): one image, bit for bit
    assert same_bits(a, ops.splat_fwd(rays, f._xs, f._ys, variant=9 - variant, cull=False))


def test_a_non_finite_cotangent_under_anomaly_mode_runs_the_backward_dense():
    """with lists, a NaN in the image cotangent gives 0 (the reference: NaN) for the rays that were dropped —
    the one documented divergence.  Under torch.autograd.set_detect_anomaly(True) render's backward checks its cotangent
    and runs dense: the dense gradient's NaN pattern, bit for bit; a finite cotangent keeps the lists (and the bits)."""
    from doodle_amd import native
    ops = native.get_ops()
    N, B, R = 600, 96, 256
    f, suns, act, _ = field_and_rays(N, B, R, 0.01, 90.0, seed=11, span=40.0)
    assert ops.lib.helio_bwd_scratch_bytes(B, N, R, 0) > 0
    G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    Gn = G.clone()
    Gn[3, 17, 40] = float("nan")

    def grad(cot, anomaly, cull=True):
        ops.cull = cull
        try:
            with torch.autograd.set_detect_anomaly(anomaly, check_nan=False):
                a = act.reshape(B, -1).clone().requires_grad_(True)
                img, _ = f.render(suns, a, None)
                (g,) = torch.autograd.grad(img, a, grad_outputs=cot)
        finally:
            ops.cull = True
        return g

    dense = grad(Gn, False, cull=False)
    nan_rows = torch.isnan(dense.view(B, N, 3)).any(dim=2)
    assert nan_rows[3].all() and not nan_rows[[0, 1, 2, 4]].any()            # 0·NaN: every ray of image 3, nothing else
    listed = grad(Gn, False)
    assert not torch.isnan(listed.view(B, N, 3)[3]).any(dim=1).all()         # the documented divergence: dropped rays read 0
    checked = grad(Gn, True)
    assert torch.equal(torch.isnan(checked), torch.isnan(dense))
    ok = ~torch.isnan(dense)
    assert torch.equal(bits(checked)[ok], bits(dense)[ok])
    assert same_bits(grad(G, True), grad(G, False)) and ops.cull             # finite: the lists, the same bits, flag restored


@pytest.mark.parametrize("N,B,R", [(96, 5, 260), (160, 3, 512), (200, 4, 128), (256, 2, 300), (65, 3, 200)])
@pytest.mark.parametrize("sigma,err", [(0.01, 90.0), (0.1, 90.0)])
def test_64_ray_tiles_walk_lists_from_65_rays(N, B, R, sigma, err):
    """Fields of 65–256 heliostats are ONE 256-ray tile (nothing to skip) but two to four 64-ray tiles: variant 12 walks lists
    there too (round 4: thousands of suns over a small field at err 90 — B=2048, N=96, R=256: 638 → 474 µs).  Culled == dense
    bit for bit, both list layouts; the size query asks for them only where the call is long."""
    from doodle_amd import native
    ops = native.get_ops()
    f, suns, act, rays = field_and_rays(N, B, R, sigma, err, seed=N + R + 5, span=30.0)
    G = torch.randn(B, R, R, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    dense = ops.splat_bwd(rays, f._xs, f._ys, G, variant=12, cull=False)
    culled, counts, need = splat_bwd_with_list(rays, f._xs, f._ys, G, 12)
    assert same_bits(culled, dense) and int(counts.min()) >= 0 and int(counts.max()) <= N
    assert same_bits(dense, ops.splat_bwd(rays, f._xs, f._ys, G, variant=2, cull=False))
    if bwd_lists_per_image(R, 12) > 1:
        fallback, counts1, _ = splat_bwd_with_list(rays, f._xs, f._ys, G, 12, per_image=True)
        assert same_bits(fallback, dense) and counts1.numel() == B
    if sigma == 0.01:
        assert int(counts.sum()) < counts.numel() * N
    else:
        assert int(counts.sum()) == counts.numel() * N
    assert ops.lib.helio_bwd_scratch_bytes(B, N, R, 12) == 0                       # a short call: dense
    assert ops.lib.helio_bwd_scratch_bytes(2048, 96, 256, 0) > 0 and ops.lib.helio_render_bwd_choice(2048, 96, 256) == 12
    assert ops.lib.helio_bwd_scratch_bytes(2048, 64, 256, 0) == 0 and ops.lib.helio_bwd_scratch_bytes(2048, 200, 256, 2) == 0
