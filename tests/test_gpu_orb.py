"""GPU parity of K5 (ORB detection per mask) and K6' (descriptors of oriented multi-level keypoints) against
the CPU oracle: keypoint coordinates, angles, levels, responses, order, counts and descriptor bytes bit-exact
(float32 arithmetic with a pinned operation order)."""
import numpy as np
import pytest
import torch

import oracle
from test_gpu_detect import _sector_masks
from test_gpu_image import _textured, _to
from vo_single_camera_sos_amd import orb_pattern

pytestmark = pytest.mark.gpu


def _contrast(rng, shape):
    import scipy.ndimage as ndi
    img = ndi.gaussian_filter(rng.random(shape) * 255, 1.1)
    return np.clip((img - img.mean()) * 6 + 128, 0, 255).astype(np.uint8)


def test_mask_pyramid_matches_oracle_resize(ctx):
    rng = np.random.default_rng(0)
    rows, cols, nmask = 122, 500, 5
    bits = np.stack([_sector_masks(rows, cols, nmask, rng), _sector_masks(rows, cols, nmask, rng, False)])
    (t_bits,) = _to(ctx.device, bits)
    pyr = ctx.orb_mask_pyramid(t_bits, nmask)
    ctx.synchronize()
    pyr = pyr.cpu().numpy()
    assert pyr.shape == (2, ctx.orb_pyramid_pixels(rows, cols))
    for s in range(2):
        planes = [((bits[s] >> m) & 1).astype(np.uint8) * 255 for m in range(nmask)]
        off, h, w = 0, rows, cols
        for l in range(8):
            if l > 0:
                h1, w1 = oracle.orb_level_size(rows, cols, l)
                planes = [(oracle.resize_linear(p, h1, w1) > 254).astype(np.uint8) * 255 for p in planes]
                h, w = h1, w1
            want = np.zeros((h, w), np.uint32)
            for m, p in enumerate(planes):
                want |= (p > 0).astype(np.uint32) << np.uint32(m)
            assert np.array_equal(pyr[s, off:off + h * w].reshape(h, w), want), (s, l)
            off += h * w


@pytest.mark.parametrize("shape,nmask,nfeatures,cap", [((122, 1200), 12, 250, 256), ((122, 600), 4, 60, 96),
                                                       ((200, 260), 1, 300, 512), ((64, 300), 2, 50, 64)])
def test_detect_orb_matches_oracle(ctx, shape, nmask, nfeatures, cap):
    rng = np.random.default_rng(shape[1] + nmask)
    NI = 4
    imgs = np.stack([_contrast(rng, shape) for _ in range(NI)])
    imgs[2] = oracle.median_gray(_textured(rng, shape + (3,)), 0)
    imgs[3] = 80
    bits = np.stack([_sector_masks(shape[0], shape[1], nmask, rng), _sector_masks(shape[0], shape[1], nmask, rng, False)])
    t_img, t_bits = _to(ctx.device, imgs, bits)
    mask_pyr = ctx.orb_mask_pyramid(t_bits, nmask)
    kp4, resp, n = ctx.detect_orb(t_img, mask_pyr, 2, nmask, nfeatures, cap)
    ctx.synchronize()
    kp4, resp, n = kp4.cpu().numpy(), resp.cpu().numpy(), n.cpu().numpy()
    total = 0
    for i in range(NI):
        want = oracle.orb_detect(imgs[i], bits[i // 2], nmask, nfeatures, cap)
        for m in range(nmask):
            p = i * nmask + m
            wk, wr = want[m]
            assert n[p] == len(wk), (i, m, n[p], len(wk))
            assert np.array_equal(kp4[p, : n[p]], wk), (i, m)
            assert np.array_equal(resp[p, : n[p]], wr), (i, m)
            total += n[p]
    assert n[3 * nmask:].sum() == 0
    if shape[0] > 62:
        assert total > 10 * nmask


def test_describe_orb_levels_matches_oracle(ctx):
    rng = np.random.default_rng(5)
    NI, nmask, cap, rows, cols = 3, 4, 128, 122, 500
    imgs = np.stack([_contrast(rng, (rows, cols)) for _ in range(NI)])
    bits = np.stack([_sector_masks(rows, cols, nmask, rng)] * 2)
    t_img, t_bits, t_pat = _to(ctx.device, imgs, bits, orb_pattern.orb_pattern())
    mask_pyr = ctx.orb_mask_pyramid(t_bits, nmask)
    kp4, resp, n = ctx.detect_orb(t_img, mask_pyr, 2, nmask, 120, cap)
    kp_in, n_in = kp4.clone(), n.clone()
    # add synthetic keypoints near the level-0 border and with arbitrary angles / levels
    extra = torch.tensor([[31.0, 31.0, 12.5, 0], [30.5, 60.0, 200.0, 0], [cols - 31.0, 60.0, 0.0, 1],
                          [100.0, rows - 31.5, 359.9, 2], [250.7, 61.2, 45.0, 3]], dtype=torch.float32, device=ctx.device)
    n_host = n_in.cpu().numpy()
    k0 = int(n_host[0])
    assert k0 + 5 <= cap
    kp_in[0, k0:k0 + 5] = extra
    n_in[0] = k0 + 5
    kp_before, n_before = kp_in.cpu().numpy(), n_in.cpu().numpy()
    desc, kp_xy = ctx.describe_orb_levels(t_img, kp_in, n_in, nmask, t_pat)
    ctx.synchronize()
    desc, kp_xy, kp_after, n_after = desc.cpu().numpy(), kp_xy.cpu().numpy(), kp_in.cpu().numpy(), n_in.cpu().numpy()
    kept_total = 0
    for p in range(NI * nmask):
        wd, kept = oracle.orb_describe_levels(imgs[p // nmask], kp_before[p, : n_before[p]])
        assert n_after[p] == len(kept), p
        assert np.array_equal(kp_after[p, : len(kept)], kp_before[p, kept]), p
        assert np.array_equal(kp_xy[p, : len(kept)], kp_before[p, kept][:, :2]), p
        assert np.array_equal(desc[p, : len(kept)], wd), p
        kept_total += len(kept)
    assert kept_total > 200


def test_describe_does_not_reuse_a_stale_pyramid(ctx):
    """detect, OVERWRITE the gray buffer in place, describe: the descriptors must come from the new images (the
    describe call builds its own pyramid; round 1 reused the detector's by pointer identity)."""
    rng = np.random.default_rng(9)
    NI, nmask, cap, rows, cols = 2, 2, 128, 122, 400
    imgs = np.stack([_contrast(rng, (rows, cols)) for _ in range(NI)])
    imgs2 = np.stack([_contrast(rng, (rows, cols)) for _ in range(NI)])
    bits = np.stack([_sector_masks(rows, cols, nmask, rng)])
    t_img, t_bits, t_pat, t_img2 = _to(ctx.device, imgs, bits, orb_pattern.orb_pattern(), imgs2)
    mask_pyr = ctx.orb_mask_pyramid(t_bits, nmask)
    kp4, resp, n = ctx.detect_orb(t_img, mask_pyr, NI, nmask, 100, cap)
    kp_before, n_before = kp4.cpu().numpy().copy(), n.cpu().numpy().copy()
    t_img.copy_(t_img2)                      # same pointer, new content
    desc, kp_xy = ctx.describe_orb_levels(t_img, kp4, n, nmask, t_pat)
    ctx.synchronize()
    desc = desc.cpu().numpy()
    differs = 0
    for p in range(NI * nmask):
        wd, kept = oracle.orb_describe_levels(imgs2[p // nmask], kp_before[p, : n_before[p]])
        assert np.array_equal(desc[p, : len(kept)], wd), p
        stale, _ = oracle.orb_describe_levels(imgs[p // nmask], kp_before[p, : n_before[p]])
        differs += int((stale != wd).any())
    assert differs > 0                       # the test can tell the two pyramids apart


@pytest.mark.parametrize("shape,nmask,nfeatures,cap", [((146, 1440), 12, 300, 384), ((122, 600), 4, 60, 96), ((64, 300), 2, 50, 64)])
def test_detect_describe_orb_equals_the_two_calls(ctx, shape, nmask, nfeatures, cap):
    """sosvo_detect_describe_orb (one pyramid, built only as far as a keypoint can come from) == detect, then describe;
    and both equal the oracle."""
    rng = np.random.default_rng(shape[1] * 3 + nmask)
    NI = 4
    imgs = np.stack([_contrast(rng, shape) for _ in range(NI)])
    imgs[1] = oracle.median_gray(_textured(rng, shape + (3,)), 0)
    bits = np.stack([_sector_masks(shape[0], shape[1], nmask, rng), _sector_masks(shape[0], shape[1], nmask, rng, False)])
    t_img, t_bits, t_pat = _to(ctx.device, imgs, bits, orb_pattern.orb_pattern())
    mask_pyr = ctx.orb_mask_pyramid(t_bits, nmask)
    kp4, resp, n = ctx.detect_orb(t_img, mask_pyr, 2, nmask, nfeatures, cap)
    det_kp, det_n = kp4.cpu().numpy().copy(), n.cpu().numpy().copy()
    desc, kp_xy = ctx.describe_orb_levels(t_img, kp4, n, nmask, t_pat)
    P = NI * nmask
    z = lambda shp, dt: torch.full(shp, 7, dtype=dt, device=ctx.device)  # noqa: E731  (dirty output buffers)
    kp4_f, resp_f, n_f = z((P, cap, 4), torch.float32), z((P, cap), torch.float32), z((P,), torch.int32)
    desc_f, kp_xy_f = z((P, cap, 32), torch.uint8), z((P, cap, 2), torch.float32)
    ctx.detect_describe_orb(t_img, mask_pyr, 2, nmask, nfeatures, t_pat, kp4_f, resp_f, n_f, desc_f, kp_xy=kp_xy_f)
    ctx.synchronize()
    assert torch.equal(n, n_f)
    nn = n.cpu().numpy()
    for p in range(P):
        k = int(nn[p])
        assert torch.equal(kp4[p, :k], kp4_f[p, :k]) and torch.equal(desc[p, :k], desc_f[p, :k]), p
        assert torch.equal(kp_xy[p, :k], kp_xy_f[p, :k]), p
        want_d, kept = oracle.orb_describe_levels(imgs[p // nmask], det_kp[p, : det_n[p]])
        assert k == len(kept) and np.array_equal(desc_f[p, :k].cpu().numpy(), want_d), p
    if shape[0] > 62:
        assert int(nn.sum()) > 5 * nmask


def test_detect_orb_on_noise_with_more_local_maxima_than_the_candidate_list(ctx):
    """A noise image has thousands of FAST local maxima per level (the on-chip candidate list holds 2048): the retainBest
    threshold still comes from ALL of them and exactly the retained ones go on (found by scripts/fuzz_parity.py)."""
    rng = np.random.default_rng(45)
    imgs = rng.integers(0, 256, (2, 147, 360), dtype=np.uint8)
    bits = np.ones((1, 147, 360), np.uint32)
    bits[0, :, 200:] |= 2
    t_img, t_bits = _to(ctx.device, imgs, bits)
    pyr = ctx.orb_mask_pyramid(t_bits, 2)
    kp4, resp, n = ctx.detect_orb(t_img, pyr, 2, 2, 230, 1024)
    ctx.synchronize()
    kp4, resp, n = kp4.cpu().numpy(), resp.cpu().numpy(), n.cpu().numpy()
    for i in range(2):
        want = oracle.orb_detect(imgs[i], bits[0], 2, 230, 1024)
        for m in range(2):
            wkp, wresp = want[m]
            p = i * 2 + m
            assert n[p] == len(wkp) and len(wkp) > 100
            assert np.array_equal(kp4[p, :n[p]], wkp) and np.array_equal(resp[p, :n[p]], wresp), (i, m)


def test_detect_describe_orb_on_an_image_too_tall_for_the_level_passes(ctx):
    """orb_level_pass_kernel keeps a level's row table in LDS (1018 rows); a taller image takes the separate resize / FAST /
    blur kernels.  Both routes must give the oracle's keypoints and descriptors."""
    rng = np.random.default_rng(77)
    shape, nmask, nfeatures, cap = (1100, 150), 2, 120, 256
    imgs = np.stack([_contrast(rng, shape) for _ in range(2)])
    bits = _sector_masks(shape[0], shape[1], nmask, rng)[None]
    t_img, t_bits, t_pat = _to(ctx.device, imgs, bits, orb_pattern.orb_pattern())
    mask_pyr = ctx.orb_mask_pyramid(t_bits, nmask)
    P = 2 * nmask
    z = lambda shp, dt: torch.full(shp, 7, dtype=dt, device=ctx.device)  # noqa: E731
    kp4, resp, n = z((P, cap, 4), torch.float32), z((P, cap), torch.float32), z((P,), torch.int32)
    desc, kp_xy = z((P, cap, 32), torch.uint8), z((P, cap, 2), torch.float32)
    ctx.detect_describe_orb(t_img, mask_pyr, 2, nmask, nfeatures, t_pat, kp4, resp, n, desc, kp_xy=kp_xy)
    ctx.synchronize()
    nn = n.cpu().numpy()
    for i in range(2):
        want = oracle.orb_detect(imgs[i], bits[0], nmask, nfeatures, cap)
        for m in range(nmask):
            p = i * nmask + m
            wk, wr = want[m]
            want_d, kept = oracle.orb_describe_levels(imgs[i], wk)
            assert nn[p] == len(kept) and len(kept) > 20, (p, nn[p], len(kept))
            assert np.array_equal(kp4[p, : nn[p]].cpu().numpy(), wk[kept]), p
            assert np.array_equal(desc[p, : nn[p]].cpu().numpy(), want_d), p


def test_detect_describe_orb_with_a_pattern_that_reaches_the_image_border(ctx):
    """A test pattern with coordinates up to +-24 reaches further than the detector's 31-px border leaves room for on a level:
    descriptors then read mirrored pixels of the blurred levels -- in the level passes the blur's mirrored columns are the
    halo lanes of the two edge strips, its mirrored rows the walk's first and last three."""
    rng = np.random.default_rng(78)
    shape, nmask, nfeatures, cap = (130, 420), 3, 150, 256
    imgs = np.stack([_contrast(rng, shape) for _ in range(2)])
    bits = _sector_masks(shape[0], shape[1], nmask, rng)[None]
    pat = rng.integers(-24, 25, size=(512, 2)).astype(np.int8)
    t_img, t_bits, t_pat = _to(ctx.device, imgs, bits, pat)
    mask_pyr = ctx.orb_mask_pyramid(t_bits, nmask)
    P = 2 * nmask
    z = lambda shp, dt: torch.full(shp, 7, dtype=dt, device=ctx.device)  # noqa: E731
    kp4, resp, n = z((P, cap, 4), torch.float32), z((P, cap), torch.float32), z((P,), torch.int32)
    desc, kp_xy = z((P, cap, 32), torch.uint8), z((P, cap, 2), torch.float32)
    ctx.detect_describe_orb(t_img, mask_pyr, 2, nmask, nfeatures, t_pat, kp4, resp, n, desc, kp_xy=kp_xy)
    ctx.synchronize()
    nn = n.cpu().numpy()
    total = 0
    for i in range(2):
        want = oracle.orb_detect(imgs[i], bits[0], nmask, nfeatures, cap)
        for m in range(nmask):
            p = i * nmask + m
            wk, wr = want[m]
            want_d, kept = oracle.orb_describe_levels(imgs[i], wk, pat)
            assert nn[p] == len(kept), (p, nn[p], len(kept))
            assert np.array_equal(desc[p, : nn[p]].cpu().numpy(), want_d), p
            total += len(kept)
    assert total > 100


def test_detect_orb_with_a_mask_that_has_no_row_inside_the_border(ctx):
    """A mask whose pixels all lie within 31 px of the top edge has columns but no rows a keypoint could come from: the
    selection must neither stage a region for it nor read through an empty row range (a GPU memory fault in round 4's first
    region-staging form, found by scripts/fuzz_parity.py)."""
    rng = np.random.default_rng(91)
    shape = (120, 300)
    imgs = np.stack([_contrast(rng, shape) for _ in range(2)])
    bits = np.zeros((1,) + shape, np.uint32)
    bits[0, :25, 40:260] |= 1          # mask 0: only border rows
    bits[0, :, 100:220] |= 2           # mask 1: a normal sector
    bits[0, 50:70, :20] |= 4           # mask 2: only border columns
    t_img, t_bits = _to(ctx.device, imgs, bits)
    pyr = ctx.orb_mask_pyramid(t_bits, 3)
    kp4, resp, n = ctx.detect_orb(t_img, pyr, 2, 3, 150, 256)
    ctx.synchronize()
    kp4, resp, n = kp4.cpu().numpy(), resp.cpu().numpy(), n.cpu().numpy()
    for i in range(2):
        want = oracle.orb_detect(imgs[i], bits[0], 3, 150, 256)
        for m in range(3):
            p = i * 3 + m
            wk, wr = want[m]
            assert n[p] == len(wk), (i, m, n[p], len(wk))
            assert np.array_equal(kp4[p, : n[p]], wk) and np.array_equal(resp[p, : n[p]], wr), (i, m)
        assert n[i * 3] == 0 and n[i * 3 + 2] == 0 and n[i * 3 + 1] > 10


def test_detect_describe_orb_at_the_largest_keypoint_capacity(ctx):
    """cap = 2048 (the ABI's limit for descriptors): the descriptor kernel's LDS is then 48 KB of keypoints beside its 24 KB
    staging area."""
    rng = np.random.default_rng(93)
    shape, nmask, nfeatures, cap = (146, 700), 1, 2000, 2048
    imgs = np.stack([_contrast(rng, shape)])
    bits = np.ones((1,) + shape, np.uint32)
    t_img, t_bits, t_pat = _to(ctx.device, imgs, bits, orb_pattern.orb_pattern())
    mask_pyr = ctx.orb_mask_pyramid(t_bits, nmask)
    z = lambda shp, dt: torch.full(shp, 7, dtype=dt, device=ctx.device)  # noqa: E731
    kp4, resp, n = z((1, cap, 4), torch.float32), z((1, cap), torch.float32), z((1,), torch.int32)
    desc, kp_xy = z((1, cap, 32), torch.uint8), z((1, cap, 2), torch.float32)
    ctx.detect_describe_orb(t_img, mask_pyr, 1, nmask, nfeatures, t_pat, kp4, resp, n, desc, kp_xy=kp_xy)
    ctx.synchronize()
    k = int(n.cpu().numpy()[0])
    wk, wr = oracle.orb_detect(imgs[0], bits[0], nmask, nfeatures, cap)[0]
    want_d, kept = oracle.orb_describe_levels(imgs[0], wk)
    assert k == len(kept) and k > 500, (k, len(kept))
    assert np.array_equal(kp4[0, :k].cpu().numpy(), wk[kept]) and np.array_equal(desc[0, :k].cpu().numpy(), want_d)
