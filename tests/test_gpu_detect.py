"""GPU parity of K4 (goodFeaturesToTrack per mask) and K6 (ORB descriptors on given keypoints) against the
CPU oracle: keypoint lists (coordinates, order, counts) and descriptor bytes bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from test_gpu_image import _textured, _to
from vo_single_camera_sos_amd import orb_pattern

pytestmark = pytest.mark.gpu


def _sector_masks(rows, cols, nmask, rng, shared_columns=True):
    """nmask azimuthal sectors as a uint32 bit field; neighbouring sectors share their boundary column (as the
    reference's rectangles do, panorama.py:573), some rows/columns are masked out everywhere."""
    bits = np.zeros((rows, cols), dtype=np.uint32)
    edges = np.linspace(0, cols - 1, nmask + 1).astype(int)
    for m in range(nmask):
        lo, hi = edges[m], edges[m + 1] if shared_columns else edges[m + 1] - 1
        bits[:, lo:hi + 1] |= np.uint32(1 << m)
    bits[:3] = 0
    bits[-4:] = 0
    bits[:, cols // 3: cols // 3 + 7] = 0   # a "stand" mask
    return bits


def _oracle_detect(gray, bits, nmask, max_corners, cap):
    eig = oracle.min_eigen(gray)
    out = []
    for m in range(nmask):
        kp, _ = oracle.gft_select(eig, bits, m, 0.01, 5.0, max_corners)
        out.append(kp[:cap])
    return out


@pytest.mark.parametrize("shape,nmask,max_corners,cap", [((122, 1200), 12, 167, 192), ((122, 1200), 12, 1000, 1024),
                                                         ((64, 150), 3, 40, 64), ((40, 70), 1, 0, 256)])
def test_detect_gft_matches_oracle(ctx, shape, nmask, max_corners, cap):
    rng = np.random.default_rng(shape[1] + nmask + max_corners)
    NI = 4
    imgs = np.stack([oracle.median_gray(_textured(rng, shape + (3,)), 0) for _ in range(NI)])
    imgs[3] = 17  # flat image: no response anywhere -> no corners
    bits = np.stack([_sector_masks(shape[0], shape[1], nmask, rng), _sector_masks(shape[0], shape[1], nmask, rng, False)])
    t_img, t_bits = _to(ctx.device, imgs, bits)
    kp, n, status = ctx.detect_gft(t_img, t_bits, 2, nmask, cap, max_corners=max_corners)
    ctx.synchronize()
    kp, n, status = kp.cpu().numpy(), n.cpu().numpy(), status.cpu().numpy()
    assert not status.any()
    total = 0
    for i in range(NI):
        want = _oracle_detect(imgs[i], bits[i // 2], nmask, max_corners, cap)
        for m in range(nmask):
            p = i * nmask + m
            assert n[p] == len(want[m]), (i, m, n[p], len(want[m]))
            assert np.array_equal(kp[p, : n[p]], want[m]), (i, m)
            total += n[p]
    assert n[3 * nmask:].sum() == 0 and total > 20 * nmask


@pytest.mark.parametrize("min_distance", [0.0, 0.5, 1.0, 1.4, 2.5, 7.5, 10.0, 33.3, 3000.0])
@pytest.mark.parametrize("max_corners", [0, 37, 300])
def test_detect_gft_minimum_distances_and_budgets(ctx, min_distance, max_corners):
    """The greedy pass on its integer grid form (fractional distances: d2 < md^2 <=> d2 < ceil(md^2); cell size 1; cells of
    several pixels), on its one-wave float form (cells beyond 2048 pixels), without a distance at all, with a budget that
    ends the pass inside a round -- raw noise (thousands of candidates, many conflicts inside a round of 64), three masks
    (256-thread workgroups) and a whole-image mask with capacity 2048 (the 1024-thread variant)."""
    rng = np.random.default_rng(int(min_distance * 10) + max_corners)
    for shape, nmask, cap in (((70, 180), 3, 512), ((96, 200), 1, 2048)):
        imgs = rng.integers(0, 256, size=(2,) + shape, dtype=np.uint8)
        imgs[1] = oracle.median_gray(_textured(rng, shape + (3,)), 0)
        bits = _sector_masks(shape[0], shape[1], nmask, rng)[None]
        t_img, t_bits = _to(ctx.device, imgs, bits)
        kp, n, status = ctx.detect_gft(t_img, t_bits, 2, nmask, cap, quality=0.002, min_distance=min_distance, max_corners=max_corners)
        ctx.synchronize()
        kp, n, status = kp.cpu().numpy(), n.cpu().numpy(), status.cpu().numpy()
        for i in range(2):
            eig = oracle.min_eigen(imgs[i])
            for m in range(nmask):
                want, _ = oracle.gft_select(eig, bits[0], m, 0.002, min_distance, max_corners)
                p = i * nmask + m
                if len(want) > cap or status[p]:   # more candidates than the workgroup sorts / corners than the capacity: flagged
                    assert status[p] != 0 or n[p] == cap, (shape, i, m, status[p])
                    continue
                assert n[p] == len(want), (shape, i, m, n[p], len(want))
                assert np.array_equal(kp[p, : n[p]], want), (shape, i, m)


def test_detect_gft_on_an_image_wider_than_the_integer_grid_form(ctx):
    """cols > 32768: the packed 16-bit distances of the grid form do not apply, the float form runs."""
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, size=(1, 12, 40000), dtype=np.uint8)
    bits = np.ones((1, 12, 40000), np.uint32)
    t_img, t_bits = _to(ctx.device, img, bits)
    kp, n, status = ctx.detect_gft(t_img, t_bits, 1, 1, 1024, quality=0.5, min_distance=9.0, max_corners=500)
    ctx.synchronize()
    want, _ = oracle.gft_select(oracle.min_eigen(img[0]), bits[0], 0, 0.5, 9.0, 500)
    assert status.item() == 0 and n.item() == len(want) and len(want) > 100
    assert np.array_equal(kp[0, : len(want)].cpu().numpy(), want)


@pytest.mark.parametrize("cols,x_off,nimg,nmask", [(1000, 0, 1, 1), (33600, 32600, 1, 1), (1000, 0, 33, 32)])
def test_detect_gft_three_corners_in_one_grid_cell(ctx, cols, x_off, nimg, nmask):
    """From a minimum distance of ~30 pixels on, a cell of round(minDistance) pixels can hold THREE accepted corners (a
    triangle 40.3 / 40.3 / 41.0 pixels at distance 40): the third must still suppress its neighbours in LATER rounds of 64
    candidates (the on-chip grid has two slots per cell and an overflow list).  Both forms of the greedy pass: integer grid,
    float one-wave (cols > 32768; or more than 1024 problems in the launch: four-wave workgroups)."""
    img = np.zeros((1, 330, cols), np.uint8)

    def blob(x, y, v):
        img[0, y - 1:y + 2, x_off + x - 1:x_off + x + 2] = v
    for k, (ox, oy) in enumerate(((0, 0), (120, 160), (240, 0))):    # three triangles, each inside one cell of 40 x 40
        for j, (x, y) in enumerate(((40, 40), (79, 50), (50, 79))):
            blob(ox + x, oy + y, 250 - 3 * (3 * k + j))
        blob(ox + 100, oy + 45, 120 - k)                             # 21.6 px from the triangle's second corner only
    for i in range(90):                                              # > 64 candidates in between: the fourth blobs come in round 2+
        blob(420 + 12 * (i % 45), 100 + 100 * (i // 45), 200)
    bits = np.full(img.shape, (1 << nmask) - 1, np.uint32)      # every mask = the whole image
    want, _ = oracle.gft_select(oracle.min_eigen(img[0]), bits[0], 0, 0.01, 40.0, 0)
    got_tri = {(int(x) - x_off, int(y)) for x, y in want}
    assert {(40, 40), (79, 50), (50, 79), (160, 200), (199, 210), (170, 239)} <= got_tri and (100, 45) not in got_tri
    t_img, t_bits = _to(ctx.device, np.repeat(img, nimg, axis=0), bits)
    cap = 256 if cols < 32768 else 2048     # (the wide image's cell grid only fits the whole-image variant's LDS)
    kp, n, status = ctx.detect_gft(t_img, t_bits, nimg, nmask, cap, min_distance=40.0, max_corners=0)
    ctx.synchronize()
    assert not status.any().item() and (n == len(want)).all().item()
    assert (kp[:, :len(want)].cpu().numpy() == want[None]).all()


def test_min_distance_and_ties_on_synthetic_response(ctx):
    """Plateaus of equal gray values give exactly tied responses: the order must be 'higher address first'."""
    img = np.zeros((60, 90), np.uint8)
    for y0 in (10, 30):
        for x0 in (10, 22, 40, 46, 70):
            img[y0:y0 + 6, x0:x0 + 6] = 200      # identical squares -> identical corner responses
    bits = np.ones((1, 60, 90), np.uint32)
    t_img, t_bits = _to(ctx.device, img[None], bits)
    kp, n, _ = ctx.detect_gft(t_img, t_bits, 1, 1, 256, max_corners=0)
    ctx.synchronize()
    want, _ = oracle.gft_select(oracle.min_eigen(img), bits[0], 0, 0.01, 5.0, 0)
    assert n.item() == len(want) and len(want) >= 20
    assert np.array_equal(kp[0, : len(want)].cpu().numpy(), want)


def test_product_pattern_equals_oracle_pattern():
    assert np.array_equal(orb_pattern.orb_pattern(), oracle.orb_pattern())


@pytest.mark.parametrize("angle", [orb_pattern.GFT_KEYPOINT_ANGLE, 0.0, 37.5, 180.0])
def test_describe_orb_matches_oracle(ctx, angle):
    rng = np.random.default_rng(int(angle * 10) + 77)
    NI, nmask, cap, rows, cols = 3, 4, 128, 122, 400
    imgs = np.stack([oracle.median_gray(_textured(rng, (rows, cols, 3)), 0) for _ in range(NI)])
    kp = np.zeros((NI * nmask, cap, 2), np.float32)
    n = rng.integers(0, cap + 1, NI * nmask).astype(np.int32)
    n[0], n[1] = 0, cap
    for p in range(NI * nmask):
        kp[p, : n[p], 0] = rng.uniform(0, cols, n[p]).round() if p % 2 else rng.uniform(0, cols, n[p])
        kp[p, : n[p], 1] = rng.uniform(0, rows, n[p]).round() if p % 2 else rng.uniform(0, rows, n[p])
    kp[2, :4] = [[31.0, 31.0], [30.99, 60.0], [cols - 31.0, 60.0], [cols - 31.01, 90.99]]
    pat = orb_pattern.orb_pattern()
    ca, sa = orb_pattern.angle_cos_sin(angle)
    t_img, t_kp, t_n, t_pat = _to(ctx.device, imgs, kp, n, pat)
    desc = ctx.describe_orb(t_img, t_kp, t_n, nmask, t_pat, ca, sa)
    ctx.synchronize()
    desc, kp_out, n_out = desc.cpu().numpy(), t_kp.cpu().numpy(), t_n.cpu().numpy()
    kept_total = 0
    for p in range(NI * nmask):
        blurred = oracle.gauss7(imgs[p // nmask])
        wd, kept = oracle.orb_describe(blurred, kp[p, : n[p]], ca, sa, pat)
        assert n_out[p] == len(kept), p
        assert np.array_equal(kp_out[p, : len(kept)], kp[p, kept]), p
        assert np.array_equal(desc[p, : len(kept)], wd), p
        kept_total += len(kept)
    assert kept_total > 100


def test_describe_orb_rows_removes_keypoints_whose_patch_leaves_the_blurred_rows(ctx):
    """sosvo_describe_orb_rows blurs only the rows of row_range; a keypoint whose patch reaches outside the trustworthy part
    (e.g. a FAST / AGAST corner of another mask set) must not read stale scratch: it is REMOVED like a keypoint within
    `edge` of the border, and every other keypoint's descriptor equals the all-rows call bit for bit.  The scratch is
    poisoned with 0xFF first, so a read of a row that was not blurred in this call would show."""
    rng = np.random.default_rng(404)
    NI, nmask, cap, rows, cols = 2, 2, 96, 146, 300
    imgs = np.stack([oracle.median_gray(_textured(rng, (rows, cols, 3)), 0) for _ in range(NI)])
    pat = orb_pattern.orb_pattern()
    ca, sa = orb_pattern.angle_cos_sin(orb_pattern.GFT_KEYPOINT_ANGLE)
    R = int(max(np.abs(np.rint(pat[:, 0] * ca - pat[:, 1] * sa)).max(), np.abs(np.rint(pat[:, 0] * sa + pat[:, 1] * ca)).max()))
    rr = np.array([[40, 120], [0, rows]], dtype=np.int32)          # view 0: an inner range; view 1: all rows
    kp = np.zeros((NI * nmask, cap, 2), np.float32)
    n = np.full(NI * nmask, cap, dtype=np.int32)
    for p in range(NI * nmask):
        kp[p, :, 0] = rng.integers(31, cols - 31, cap)
        kp[p, :, 1] = rng.integers(31, rows - 31, cap)
    kp[0, :6, 1] = [40 + 3 + R, 40 + 3 + R - 1, 120 - 3 - 1 - R, 120 - 3 - R, 35, 130]   # edge cases of view 0
    t_img, t_pat, t_rr = _to(ctx.device, imgs, pat, rr)
    t_kp_all, t_n_all = _to(ctx.device, kp.copy(), n.copy())
    d_all = ctx.describe_orb(t_img, t_kp_all, t_n_all, nmask, t_pat, ca, sa)
    ctx.synchronize()
    d_all, kp_all, n_all = d_all.cpu().numpy(), t_kp_all.cpu().numpy(), t_n_all.cpu().numpy()
    ctx.debug_fill_scratch(0xFF)
    t_kp, t_n = _to(ctx.device, kp.copy(), n.copy())
    d = ctx.describe_orb(t_img, t_kp, t_n, nmask, t_pat, ca, sa, row_range=t_rr)
    ctx.synchronize()
    d, kp_o, n_o = d.cpu().numpy(), t_kp.cpu().numpy(), t_n.cpu().numpy()
    for p in range(NI * nmask):
        view = p // nmask   # NI = 2: image i is view i
        lo, hi = rr[view]
        y = kp_all[p, : n_all[p], 1].astype(int)
        ok = (y - R >= (0 if lo == 0 else lo + 3)) & (y + R < (rows if hi == rows else hi - 3))
        assert n_o[p] == ok.sum(), (p, n_o[p], ok.sum())
        assert np.array_equal(kp_o[p, : n_o[p]], kp_all[p, : n_all[p]][ok])
        assert np.array_equal(d[p, : n_o[p]], d_all[p, : n_all[p]][ok])
    assert n_o[0] < n_all[0] and (n_o[2:] == n_all[2:]).all()
    ys = kp_o[0, : n_o[0], 1]
    assert (40 + 3 + R) in ys and (120 - 3 - 1 - R) in ys and (40 + 3 + R - 1) not in ys and (120 - 3 - R) not in ys


def test_fast_detector_raster_order_masks_and_cap(ctx):
    """sosvo_detect_fast against the oracle's FAST score map + numpy NMS / mask / raster order; overlapping masks, an
    empty mask, a cap smaller than the corner count (first `cap` in raster order, status 1)."""
    import refflow
    rng = np.random.default_rng(77)
    rows, cols = 61, 203
    imgs = np.stack([np.clip(rng.normal(120, 40, (rows, cols)), 0, 255).astype(np.uint8) for _ in range(3)])
    imgs[2] = 90                                                     # flat image: no corners
    masks = np.zeros((1, rows, cols), np.uint32)
    masks[0, :, :120] |= 1
    masks[0, 10:50, 100:] |= 2                                       # overlaps mask 0 on columns 100..119
    masks[0, :, :] |= 8                                              # mask 3 = everything (mask 2 stays empty)
    dev = ctx.device
    t_img, t_mask = torch.from_numpy(imgs).to(dev), torch.from_numpy(masks).to(dev)
    for cap in (2048, 40):
        kp, n, status = ctx.detect_fast(t_img, t_mask, 3, 4, cap)
        ctx.synchronize()
        kp, n, status = kp.cpu().numpy(), n.cpu().numpy(), status.cpu().numpy()
        for i in range(3):
            for m in range(4):
                want = refflow.fast_keypoints(imgs[i], masks[0], m)
                p = i * 4 + m
                assert n[p] == min(len(want), cap) and status[p] == (1 if len(want) > cap else 0), (cap, i, m, n[p], len(want))
                assert np.array_equal(kp[p, : n[p]], want[:cap]), (cap, i, m)
        assert n[0] >= min(cap, 100) and n[2] == 0 and n[8:].sum() == 0


def test_agast_detector_block_maxima_masks_and_cap(ctx):
    """sosvo_detect_agast: the FAST-9/16 corner set with AGAST's block-maximum suppression (oracle.agast_nms), masks,
    raster order, cap / status; a larger image with long runs of touching corners."""
    import refflow
    rng = np.random.default_rng(78)
    for rows, cols in ((61, 203), (146, 700)):
        sigma = 40 if rows < 100 else 9          # (white noise at sigma 40 makes a quarter of the pixels corners)
        imgs = np.stack([np.clip(rng.normal(120, sigma, (rows, cols)), 0, 255).astype(np.uint8) for _ in range(3)])
        imgs[1, : rows // 2] = np.clip(imgs[1, : rows // 2].astype(int) // 16 * 16 + 5, 0, 255)   # plateaus: ties in the responses
        imgs[2] = 90
        masks = np.zeros((1, rows, cols), np.uint32)
        masks[0, :, : cols // 2 + 9] |= 1
        masks[0, 10: rows - 9, cols // 2:] |= 2
        masks[0, :, :] |= 8
        t_img, t_mask = torch.from_numpy(imgs).to(ctx.device), torch.from_numpy(masks).to(ctx.device)
        for cap in (4096, 40):
            kp, n, status = ctx.detect_agast(t_img, t_mask, 3, 4, cap)
            ctx.synchronize()
            kp, n, status = kp.cpu().numpy(), n.cpu().numpy(), status.cpu().numpy()
            for i in range(3):
                for m in range(4):
                    want = refflow.agast_keypoints(imgs[i], masks[0], m)
                    p = i * 4 + m
                    assert n[p] == min(len(want), cap) and status[p] == (1 if len(want) > cap else 0), (cap, i, m, n[p], len(want))
                    assert np.array_equal(kp[p, : n[p]], want[:cap]), (cap, i, m)
            assert n[0] >= min(cap, 100) and n[2] == 0 and n[8:].sum() == 0


def test_detect_gft_reports_candidate_overflow(ctx):
    """More 3x3 local maxima above the threshold than the selection holds (4096 for cap <= 1024): status bit 0 is set
    (include/sosvo.h: "the result then depends on which were kept"); the call still returns cap corners that respect the
    minimum distance, and the large-mask variant (cap > 1024: 16384 candidates) handles the same image without the flag
    and equals the oracle."""
    rng = np.random.default_rng(77)
    rows, cols = 200, 420
    img = rng.integers(0, 256, (1, rows, cols), dtype=np.uint8)          # white noise: a local maximum every ~9 pixels
    bits = np.ones((1, rows, cols), dtype=np.uint32)
    t_img, t_bits = _to(ctx.device, img, bits)
    kp, n, status = ctx.detect_gft(t_img, t_bits, 1, 1, 512, quality=1e-4, max_corners=0)
    ctx.synchronize()
    kp, n, status = kp.cpu().numpy(), n.cpu().numpy(), status.cpu().numpy()
    assert status[0] & 1 and n[0] == 512
    d = np.linalg.norm(kp[0, :512, None, :] - kp[0, None, :512, :], axis=-1) + 1e9 * np.eye(512)
    assert d.min() >= 5.0
    kp2, n2, status2 = ctx.detect_gft(t_img, t_bits, 1, 1, 2048, quality=1e-4, max_corners=0)
    ctx.synchronize()
    assert not (status2.cpu().numpy()[0] & 1)
    want, _ = oracle.gft_select(oracle.min_eigen(img[0]), bits[0], 0, 1e-4, 5.0, 0)
    m = min(2048, len(want))
    assert n2.cpu().numpy()[0] == m and np.array_equal(kp2.cpu().numpy()[0, :m], want[:m])


def test_detect_gft_on_a_patch_where_every_pixel_is_a_maximum(ctx):
    """An exactly periodic texture gives a plateau of EQUAL positive responses: every pixel of the patch equals its 3x3
    dilation and is a candidate (four times the density distinct values allow).  A 40 x 60 patch (2400 candidates inside one
    58-column strip) must still come out exactly as the oracle's goodFeaturesToTrack (the candidate records of a wave have
    room for all of its pixels).  A frame covered by the texture has more candidates than the selection sorts (4096):
    status bit 0, call still well-formed."""
    tile = np.array([[255, 0, 0, 0], [255, 0, 0, 0], [255, 255, 0, 255]], dtype=np.uint8)
    rows, cols = 120, 256
    rng = np.random.default_rng(5)
    img = np.full((2, rows, cols), 90, dtype=np.uint8)
    img[0, 30:70, 66:126] = np.tile(tile, (14, 15))[:40, :60]
    img[0, 80:110, 150:240] = rng.integers(0, 256, (30, 90), dtype=np.uint8)      # ordinary texture next to it
    img[1] = np.tile(tile, (rows // 3 + 1, cols // 4 + 1))[:rows, :cols]
    bits = np.ones((1, rows, cols), dtype=np.uint32)
    e = oracle.min_eigen(img[0])
    assert (e[35:65, 72:120] == e[40, 80]).all() and e[40, 80] > 0                # the plateau the test is about
    t_img, t_bits = _to(ctx.device, img, bits)
    kp, n, status = ctx.detect_gft(t_img, t_bits, 2, 1, 1024, quality=0.01, max_corners=0)
    ctx.synchronize()
    kp, n, status = kp.cpu().numpy(), n.cpu().numpy(), status.cpu().numpy()
    want, _ = oracle.gft_select(e, bits[0], 0, 0.01, 5.0, 0)
    assert status[0] == 0 and n[0] == len(want) and np.array_equal(kp[0, : n[0]], want)
    assert status[1] & 1 and 0 < n[1] <= 1024
    d = np.linalg.norm(kp[1, : n[1], None, :] - kp[1, None, : n[1], :], axis=-1) + 1e9 * np.eye(n[1])
    assert d.min() >= 5.0


def test_detect_gft_accepts_quality_levels_of_one_and_more(ctx):
    """cv2.goodFeaturesToTrack takes any positive qualityLevel; at >= 1 the threshold reaches the maximum and no corner is
    left (oracle: same).  The C ABI must not refuse the call (the reference passes the parameter through)."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (1, 96, 160), dtype=np.uint8)
    bits = np.ones((1, 96, 160), dtype=np.uint32)
    t_img, t_bits = _to(ctx.device, img, bits)
    for q in (1.0, 1.5):
        kp, n, status = ctx.detect_gft(t_img, t_bits, 1, 1, 256, quality=q, max_corners=0)
        ctx.synchronize()
        want, _ = oracle.gft_select(oracle.min_eigen(img[0]), bits[0], 0, q, 5.0, 0)
        assert len(want) == 0 and n.cpu().numpy()[0] == 0 and status.cpu().numpy()[0] == 0
    kp, n, status = ctx.detect_gft(t_img, t_bits, 1, 1, 256, quality=0.999, max_corners=0)
    ctx.synchronize()
    want, _ = oracle.gft_select(oracle.min_eigen(img[0]), bits[0], 0, 0.999, 5.0, 0)
    assert n.cpu().numpy()[0] == len(want) == 1 and np.array_equal(kp.cpu().numpy()[0, :1], want)
