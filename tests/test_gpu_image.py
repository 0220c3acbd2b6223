"""GPU parity of the image stages (through the C ABI) against the CPU oracle -- all integer/byte work,
compared bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
from vo_single_camera_sos_amd.omnistereo.panorama import Panorama

pytestmark = pytest.mark.gpu


def _to(dev, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


def _textured(rng, shape):
    """Smooth random texture + sharp blobs, uint8."""
    import scipy.ndimage as ndi
    base = ndi.gaussian_filter(rng.random(shape[:2]) * 255, 2.0)
    img = np.stack([np.clip(base + rng.normal(0, 12, shape[:2]) + 20 * c, 0, 255) for c in range(3)], axis=-1)
    for _ in range(60):
        y, x = rng.integers(0, shape[0]), rng.integers(0, shape[1])
        img[max(y - 3, 0):y + 3, max(x - 3, 0):x + 3] = rng.integers(0, 256, 3)
    return img.astype(np.uint8)


@pytest.fixture(scope="module")
def rig():
    gs = synthetic_gums()
    gs.top_model.panorama = Panorama(gs.top_model, width=1200)
    gs.bot_model.panorama = Panorama(gs.bot_model, width=1200)
    gs.make_annulus_masks((480, 640))
    return gs


def test_unwrap_c2_both_views_masked(ctx, rig):
    rng = np.random.default_rng(0)
    omni = np.stack([_textured(rng, (480, 640, 3)) for _ in range(3)])
    maps = [m.panorama.float32_maps() for m in (rig.top_model, rig.bot_model)]
    mx, my = np.stack([maps[0][0], maps[1][0]]), np.stack([maps[0][1], maps[1][1]])
    masks = np.stack([rig.top_model.mask, rig.bot_model.mask])
    t_omni, t_masks, t_mx, t_my = _to(ctx.device, omni, masks, mx, my)
    pano = ctx.unwrap(t_omni, t_masks, t_mx, t_my)
    pano_nomask = ctx.unwrap(t_omni, None, t_mx, t_my)
    ctx.synchronize()
    pano, pano_nomask = pano.cpu().numpy(), pano_nomask.cpu().numpy()
    assert pano.shape == (2, 3, 122, 1200, 3)
    for v in range(2):
        for f in range(3):
            assert np.array_equal(pano[v, f], oracle.unwrap(omni[f], masks[v], mx[v], my[v])), (v, f)
            assert np.array_equal(pano_nomask[v, f], oracle.unwrap(omni[f], None, mx[v], my[v])), (v, f)
    assert pano[0].any() and pano[1].any()
    assert not pano[0, :, 0].any()   # row 0 of the top view is outside the top mirror's elevation range (NaN LUT)


def test_unwrap_random_maps_border_and_nan(ctx):
    rng = np.random.default_rng(1)
    omni = rng.integers(0, 256, (2, 37, 53, 3), dtype=np.uint8)
    mx = rng.uniform(-6, 59, (2, 40, 70)).astype(np.float32)
    my = rng.uniform(-6, 43, (2, 40, 70)).astype(np.float32)
    mx[0, 3, 5:9] = np.nan
    my[1, 7, 1] = np.nan
    mx[0, 0, :4] = [0.0, 52.0, 52.5, 53.0]
    my[0, 0, :4] = [0.0, 36.0, 36.515625, 0.015625]     # exact 1/64 fractions: round-half-even cases
    mx[1, 1, :2] = [10.015625, 10.046875]
    masks = (rng.random((2, 37, 53)) < 0.7).astype(np.uint8) * 255
    t = _to(ctx.device, omni, masks, mx, my)
    pano = ctx.unwrap(*t)
    ctx.synchronize()
    pano = pano.cpu().numpy()
    for v in range(2):
        for f in range(2):
            assert np.array_equal(pano[v, f], oracle.unwrap(omni[f], masks[v], mx[v], my[v]))


@pytest.mark.parametrize("pshape", [(40, 72), (41, 71), (1, 5)])
def test_unwrap_table_form_equals_map_form(ctx, pshape):
    """The batched path's table-driven K1 (sosvo_unwrap_prepare + sosvo_unwrap_table) is bit-identical to the
    map-driven one (and hence to the oracle), also for panorama sizes that defeat the aligned 4-pixel stores."""
    rng = np.random.default_rng(pshape[1])
    omni = rng.integers(0, 256, (3, 37, 53, 3), dtype=np.uint8)
    mx = rng.uniform(-6, 59, (2,) + pshape).astype(np.float32)
    my = rng.uniform(-6, 43, (2,) + pshape).astype(np.float32)
    mx[0, 0, :3] = [np.nan, 52.0, 51.99]
    my[0, 0, :3] = [3.0, 36.0, 35.99]       # the very last source pixel as a tap
    masks = (rng.random((2, 37, 53)) < 0.7).astype(np.uint8) * 255
    masks[:, 36, 52] = 255
    t_omni, t_masks, t_mx, t_my = _to(ctx.device, omni, masks, mx, my)
    for tm in (t_masks, None):
        want = ctx.unwrap(t_omni, tm, t_mx, t_my)
        got = ctx.unwrap_table(t_omni, ctx.unwrap_prepare(tm, t_mx, t_my, (37, 53)))
        ctx.synchronize()
        assert torch.equal(got, want)
    assert np.array_equal(got[0, 2].cpu().numpy(), oracle.unwrap(omni[2], None, mx[0], my[0]))


@pytest.mark.parametrize("shape,k", [((122, 1200), 11), ((37, 53), 11), ((64, 54), 11), ((30, 109), 5),
                                     ((20, 63), 3), ((11, 11), 11), ((5, 200), 11)])
def test_median_gray_exact(ctx, shape, k):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1] + k)
    imgs = np.stack([_textured(rng, shape + (3,)), rng.integers(0, 256, shape + (3,), dtype=np.uint8),
                     np.full(shape + (3,), 255, np.uint8), np.zeros(shape + (3,), np.uint8)])
    imgs[1, : shape[0] // 2] //= 64          # few distinct values: many ties inside the window
    (t,) = _to(ctx.device, imgs)
    gray = ctx.median_gray(t, k)
    ctx.synchronize()
    gray = gray.cpu().numpy()
    for i in range(imgs.shape[0]):
        assert np.array_equal(gray[i], oracle.median_gray(imgs[i], k)), i


def test_gray_only(ctx):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (3, 48, 64, 3), dtype=np.uint8)
    (t,) = _to(ctx.device, img)
    gray = ctx.median_gray(t, 0)
    ctx.synchronize()
    for i in range(3):
        assert np.array_equal(gray[i].cpu().numpy(), oracle.median_gray(img[i], 0))


def test_get_panoramic_image_mirror_method(ctx, rig):
    """Panorama.get_panoramic_image (the reference's per-mirror entry point) runs the same kernel."""
    rng = np.random.default_rng(6)
    omni = _textured(rng, (480, 640, 3))
    pn = rig.bot_model.panorama
    out = pn.get_panoramic_image(omni)
    mx, my = pn.float32_maps()
    assert np.array_equal(out, oracle.unwrap(omni, None, mx, my))
    assert np.array_equal(pn.panoramic_img, out)


@pytest.mark.parametrize("pshape,k", [((40, 72), 11), ((41, 71), 5), ((13, 130), 3), ((1, 5), 11), ((40, 72), 0), ((17, 301), 1)])
def test_fused_unwrap_median_gray_equals_oracle_chain(ctx, pshape, k):
    """sosvo_unwrap_median_gray (K1 inside the median kernel, no colour panorama in HBM) against
    oracle.unwrap -> oracle.median_gray; taps on the first/last bytes of a frame included."""
    rng = np.random.default_rng(pshape[1] + k)
    omni = rng.integers(0, 256, (3, 37, 53, 3), dtype=np.uint8)
    mx = rng.uniform(-6, 59, (2,) + pshape).astype(np.float32)
    my = rng.uniform(-6, 43, (2,) + pshape).astype(np.float32)
    mx[0, 0, :5] = [np.nan, 52.0, 51.99, -0.5, 0.0]
    my[0, 0, :5] = [3.0, 36.0, 35.99, 0.0, 0.0]  # last source pixel, a tap row starting before the frame, first pixel
    masks = (rng.random((2, 37, 53)) < 0.7).astype(np.uint8) * 255
    masks[:, 36, 52] = masks[:, 0, 0] = 255
    t_omni, t_masks, t_mx, t_my = _to(ctx.device, omni, masks, mx, my)
    for tm, hm in ((t_masks, masks), (None, [None, None])):
        table = ctx.unwrap_prepare(tm, t_mx, t_my, (37, 53))
        gray = ctx.unwrap_median_gray(t_omni, table, k)
        chain = ctx.median_gray(ctx.unwrap_table(t_omni, table), k)
        ctx.set_hint_shared_device(True)     # sosvo_set_hint: a scheduling hint (three median workgroups per CU), never a result
        hinted = ctx.unwrap_median_gray(t_omni, table, k)
        ctx.set_hint_shared_device(False)
        ctx.synchronize()
        assert torch.equal(gray.view_as(chain), chain) and torch.equal(hinted, gray)
        gray = gray.cpu().numpy().reshape(2, 3, *pshape)
        for v in range(2):
            for f in range(3):
                assert np.array_equal(gray[v, f], oracle.median_gray(oracle.unwrap(omni[f], hm[v], mx[v], my[v]), k))


def test_fused_front_end_c2(ctx, rig):
    """C2 geometry: the fused front end (keep_panoramas=False) produces the gray images of the unfused one."""
    from vo_single_camera_sos_amd.frontend import DeviceImageModel, ImageFrontEnd
    rng = np.random.default_rng(8)
    omni = np.stack([_textured(rng, (480, 640, 3)) for _ in range(2)])
    model = DeviceImageModel(ctx, rig, (480, 640))
    a = ImageFrontEnd(ctx, model, 2, num_of_features=50)
    b = ImageFrontEnd(ctx, model, 2, num_of_features=50, keep_panoramas=False, skip_unreachable_rows=False)
    c = ImageFrontEnd(ctx, model, 2, num_of_features=50, keep_panoramas=False)   # + rows no consumer reaches skipped
    for fe in (a, b, c):
        fe.load_frames(omni)
        fe.run()
    ctx.synchronize()
    assert b.pano is None and torch.equal(a.gray, b.gray) and torch.equal(a.kp, b.kp) and torch.equal(a.desc, b.desc)
    assert int(a.n.sum()) > 100
    # the row-restricted form: identical keypoints and descriptors; gray identical inside each view's range, untouched
    # (the buffer's zeros) outside; the ranges are what the masks, the GFT halo and the descriptor border imply
    assert torch.equal(a.kp, c.kp) and torch.equal(a.n, c.n) and torch.equal(a.desc, c.desc)
    rr = c.gray_rows.cpu().numpy()
    full, part = a.gray.cpu().numpy().reshape(2, 2, model.rows, model.cols), c.gray.cpu().numpy().reshape(2, 2, model.rows, model.cols)
    pat = model.pattern_host.astype(np.float32)
    ca, sa = np.float32(c.cos_a), np.float32(c.sin_a)
    Rp = int(max(np.abs(np.rint(pat[:, 0] * ca - pat[:, 1] * sa)).max(), np.abs(np.rint(pat[:, 0] * sa + pat[:, 1] * ca)).max()))
    for v in range(2):
        mrows = np.where((model.mask_bits_host[v] != 0).any(axis=1))[0]
        mlo, mhi = int(mrows[0]), int(mrows[-1])
        ylo, yhi = max(31, mlo), min(model.rows - 32, mhi)
        lo = max(0, min(mlo - 3, ylo - Rp - 3))
        hi = min(model.rows, max(mhi + 3, yhi + Rp + 3) + 1)
        assert (int(rr[v, 0]), int(rr[v, 1])) == (lo, hi), (v, rr[v], lo, hi)
        assert np.array_equal(part[v, :, lo:hi], full[v, :, lo:hi])
        assert not part[v, :, :lo].any() and not part[v, :, hi:].any()
    assert (rr[:, 1] - rr[:, 0]).sum() < 2 * model.rows      # something is actually skipped


def test_unwrap_median_gray_rows_api(ctx):
    """sosvo_unwrap_median_gray_rows: inside the per-view row range the result is the full call's, outside the
    caller's buffer is left alone; an empty range writes nothing."""
    rng = np.random.default_rng(12)
    omni = rng.integers(0, 256, (3, 37, 53, 3), dtype=np.uint8)
    pshape = (40, 150)
    mx = rng.uniform(-2, 55, (2,) + pshape).astype(np.float32)
    my = rng.uniform(-2, 39, (2,) + pshape).astype(np.float32)
    t_omni, t_mx, t_my = _to(ctx.device, omni, mx, my)
    table = ctx.unwrap_prepare(None, t_mx, t_my, (37, 53))
    full = ctx.unwrap_median_gray(t_omni, table, 11)
    for rng_rows in ([[7, 29], [0, 40]], [[0, 1], [39, 40]], [[5, 5], [12, 30]], [[-3, 90], [20, 10]]):
        rr = torch.tensor(rng_rows, dtype=torch.int32, device=ctx.device)
        out = torch.full_like(full, 77)
        ctx.unwrap_median_gray(t_omni, table, 11, gray=out, row_range=rr)
        ctx.synchronize()
        f, o = full.cpu().numpy().reshape(2, 3, *pshape), out.cpu().numpy().reshape(2, 3, *pshape)
        for v in range(2):
            lo = max(0, min(pshape[0], rng_rows[v][0]))
            hi = max(lo, min(pshape[0], rng_rows[v][1]))
            assert np.array_equal(o[v, :, lo:hi], f[v, :, lo:hi]), (rng_rows, v)
            assert (o[v, :, :lo] == 77).all() and (o[v, :, hi:] == 77).all(), (rng_rows, v)
