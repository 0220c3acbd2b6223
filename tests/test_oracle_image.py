"""CPU: pin the image-stage oracle (OpenCV semantics restated; the reference holds no golden vectors for
these stages) with hand-checkable known answers and independent numpy/scipy evaluations."""
import numpy as np
import pytest
import scipy.ndimage as ndi

import oracle


def test_unwrap_identity_fraction_border_and_nan():
    rng = np.random.default_rng(0)
    omni = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    # integer coordinates reproduce the pixel; (x+0.5, y) averages two pixels with rounding
    mx = np.array([[3.0, 7.5, -1.0, 29.0, 29.5, np.nan, 12.25]], dtype=np.float32)
    my = np.array([[4.0, 2.0, 5.0, 19.0, 19.0, 3.0, 6.75]], dtype=np.float32)
    p = oracle.unwrap(omni, None, mx, my)[0]
    assert np.array_equal(p[0], omni[4, 3])
    assert np.array_equal(p[1], (omni[2, 7].astype(int) + omni[2, 8] + 1) // 2)
    assert np.array_equal(p[2], [0, 0, 0])                                  # fully outside -> border 0
    assert np.array_equal(p[3], omni[19, 29])
    assert np.array_equal(p[4], (omni[19, 29].astype(int) * 512 + 512) >> 10)  # right tap is border (0)
    assert np.array_equal(p[5], [0, 0, 0])                                  # NaN map entry
    w = np.array([24 * 8, 8 * 8, 24 * 24, 8 * 24])  # fx = 8/32, fy = 24/32
    taps = np.stack([omni[6, 12], omni[6, 13], omni[7, 12], omni[7, 13]]).astype(int)
    assert np.array_equal(p[6], ((w[:, None] * taps).sum(0) + 512) >> 10)
    # folding the annulus mask into the taps == masking the image first (bitwise_and, camera_models.py:2992)
    mask = (rng.random((20, 30)) < 0.6).astype(np.uint8) * 255
    gx, gy = np.meshgrid(np.linspace(-2, 31, 40), np.linspace(-2, 21, 25))
    gx, gy = gx.astype(np.float32), gy.astype(np.float32)
    assert np.array_equal(oracle.unwrap(omni, mask, gx, gy), oracle.unwrap(omni * (mask[..., None] > 0), None, gx, gy))


def test_median_and_gray_against_scipy():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    img[10:20, 5:30] = 200  # flat area
    gray, bgr = oracle.median_gray(img, 11, want_bgr=True)
    for c in range(3):
        assert np.array_equal(bgr[..., c], ndi.median_filter(img[..., c], size=11, mode="nearest"))
    b, g, r = [bgr[..., c].astype(np.int64) for c in range(3)]
    assert np.array_equal(gray, (1868 * b + 9617 * g + 4899 * r + 8192) >> 14)
    i64 = img.astype(np.int64)
    assert np.array_equal(oracle.median_gray(img, 0), (1868 * i64[..., 0] + 9617 * i64[..., 1] + 4899 * i64[..., 2]
                                                       + 8192) >> 14)
    assert oracle.median_gray(np.full((5, 5, 3), 255, np.uint8), 3).max() == 255


def test_min_eigen_against_float64_structure_tensor():
    rng = np.random.default_rng(2)
    g = ndi.gaussian_filter(rng.random((40, 60)) * 255, 1.5).astype(np.uint8)
    eig = oracle.min_eigen(g)
    gd = g.astype(np.float64)
    dx = ndi.correlate(gd, np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1.]]), mode="mirror") / 3060.0
    dy = ndi.correlate(gd, np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1.]]), mode="mirror") / 3060.0
    box = lambda a: ndi.uniform_filter(a, 3, mode="mirror") * 9  # noqa: E731
    a, b, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    want = (a + c) - np.sqrt((a - c) ** 2 + b * b)
    assert np.allclose(eig, want, rtol=2e-4, atol=2e-7)
    assert oracle.min_eigen(np.full((9, 9), 77, np.uint8)).max() == 0.0  # flat image: no corner response


def test_gft_selection_rules():
    eig = np.zeros((30, 40), dtype=np.float32)
    mask = np.ones((30, 40), dtype=np.uint32)       # bit 0: one mask everywhere
    eig[10, 10] = 1.0
    eig[10, 13] = 0.9                               # 3 px from the first: suppressed by min distance 5
    eig[10, 16] = 0.8                               # 6 px: kept
    eig[20, 20] = 0.006                             # below 0.01 * max (2.0), above 0.01 * 0.5
    eig[5, 30] = 0.5
    eig[5, 31] = 0.5                                # equal neighbours: both are 3x3 maxima; higher address first,
    eig[0, 5] = 2.0                                 # then the other is within min distance.  Border row: excluded,
    kp, maxv = oracle.gft_select(eig, mask, 0)      # but it still sets the maximum (and hence the threshold)
    assert maxv == 2.0
    assert kp.tolist() == [[10.0, 10.0], [16.0, 10.0], [31.0, 5.0]]
    kp2, _ = oracle.gft_select(eig, mask, 0, max_corners=2)
    assert kp2.tolist() == [[10.0, 10.0], [16.0, 10.0]]
    # a second mask sees only its own pixels for max / candidates, but dilation looks at all pixels
    mask[:, 20:] = 2                                # bit 1 only
    mask[:, 19] = 3                                 # masks may share a column
    kp3, maxv3 = oracle.gft_select(eig, mask, 1)
    assert maxv3 == 0.5 and kp3.tolist() == [[31.0, 5.0], [20.0, 20.0]]
    kp4, _ = oracle.gft_select(eig, mask, 7)
    assert kp4.shape == (0, 2)


@pytest.mark.parametrize("shape,md", [((12, 40000), 9.0), ((60, 90), 2.5), ((40, 300), 7.5)])
def test_gft_minimum_distance_equals_the_all_pairs_rule(shape, md):
    """The cell grid of the selection is an accelerator only: the corners are those of the plain rule 'accept a candidate
    iff no accepted one is closer than the distance' over the sorted candidates -- also on an image wider than 32767
    columns (the grid's coordinates were 16-bit until round 3) and with a fractional distance."""
    rng = np.random.default_rng(shape[1])
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    bits = np.ones(shape, np.uint32)
    eig = oracle.min_eigen(img)
    want, _ = oracle.gft_select(eig, bits, 0, 0.5, md, 300)
    cand, _ = oracle.gft_select(eig, bits, 0, 0.5, 0.0, 0)
    acc = []
    for x, y in cand:
        if all(np.float32((x - ax) ** 2 + (y - ay) ** 2) >= np.float32(md * md) for ax, ay in acc):
            acc.append((x, y))
            if len(acc) == 300:
                break
    assert len(want) > 20 and np.array_equal(want, np.array(acc, np.float32))


def test_gauss7_is_normalised_and_symmetric():
    flat = np.full((20, 20), 131, np.uint8)
    assert np.array_equal(oracle.gauss7(flat), flat)
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (25, 31), dtype=np.uint8)
    k = np.array([18, 34, 49, 54, 49, 34, 18], dtype=np.float64) / 256
    want = ndi.correlate1d(ndi.correlate1d(g.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    assert np.abs(oracle.gauss7(g).astype(np.float64) - want).max() <= 0.5 + 1e-9
    assert np.array_equal(oracle.gauss7(g[::-1, ::-1]), oracle.gauss7(g)[::-1, ::-1])


def test_orb_pattern_and_descriptor_bits():
    seeded = oracle.orb_pattern_seeded()
    assert seeded.shape == (512, 2) and np.abs(seeded).max() <= 12
    assert np.array_equal(seeded, oracle.orb_pattern_seeded())             # deterministic
    sp = seeded.reshape(256, 4)
    assert not np.any((sp[:, 0] == sp[:, 2]) & (sp[:, 1] == sp[:, 3]))
    # the default: OpenCV's bit_pattern_31_ (known first / last rows of that table, |coordinate| <= 13)
    pat = oracle.orb_pattern()
    pairs = pat.reshape(256, 4)
    assert pat.shape == (512, 2) and pat.dtype == np.int8 and np.abs(pat).max() == 13
    assert pairs[:4].tolist() == [[8, -3, 9, 5], [4, 2, 7, -12], [-11, 9, -8, 2], [7, -12, 12, -13]]
    assert pairs[-1].tolist() == [-1, -6, 0, -11]
    assert not np.any((pairs[:, 0] == pairs[:, 2]) & (pairs[:, 1] == pairs[:, 3]))
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (100, 120), dtype=np.uint8)
    kp = np.array([[50.0, 40.0], [30.9, 31.0], [31.0, 30.9], [88.9, 68.9], [89.0, 50.0], [60.4, 45.6]], np.float32)
    desc, kept = oracle.orb_describe(img, kp, 1.0, 0.0, pat)
    assert kept.tolist() == [0, 3, 5]                                      # border rule: 31 <= x < cols - 31
    for d, i in zip(desc, kept):
        cx, cy = int(np.rint(kp[i, 0])), int(np.rint(kp[i, 1]))
        bits = [img[cy + pairs[t, 1], cx + pairs[t, 0]] < img[cy + pairs[t, 3], cx + pairs[t, 2]] for t in range(256)]
        assert np.array_equal(np.unpackbits(d, bitorder="little"), np.array(bits, dtype=np.uint8))
    # rotation by 90 degrees: offsets (x, y) -> (-y, x)
    d90, _ = oracle.orb_describe(img, kp[:1], 0.0, 1.0, pat)
    bits = [img[40 + pairs[t, 0], 50 - pairs[t, 1]] < img[40 + pairs[t, 2], 50 - pairs[t, 3]] for t in range(256)]
    assert np.array_equal(np.unpackbits(d90[0], bitorder="little"), np.array(bits, dtype=np.uint8))
