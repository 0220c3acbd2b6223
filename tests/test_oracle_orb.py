"""CPU: pin the ORB oracle (OpenCV's ORB semantics restated; no golden vectors exist in the reference) with
hand-checkable known answers and independent brute-force evaluations of the definitions."""
import numpy as np
import scipy.ndimage as ndi

import oracle

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def test_level_sizes_and_quotas():
    assert [oracle.orb_level_size(122, 1200, l) for l in range(5)] == [(122, 1200), (102, 1000), (85, 833), (71, 694),
                                                                        (59, 579)]
    assert oracle.orb_quotas(167).tolist() == [36, 30, 25, 21, 17, 15, 12, 11]
    assert oracle.orb_quotas(500).sum() == 500


def test_resize_properties():
    flat = np.full((40, 60), 93, np.uint8)
    assert np.array_equal(oracle.resize_linear(flat, 33, 50), np.full((33, 50), 93, np.uint8))
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 60), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(img, 40, 60), img)                       # identity size
    half = oracle.resize_linear(img, 20, 30)                                            # exact 2:1 -> 2x2 box mean
    want = (img.reshape(20, 2, 30, 2).astype(np.int64).sum(axis=(1, 3)) + 2) >> 2
    assert np.abs(half.astype(np.int64) - want).max() <= 1
    ramp = np.tile(np.arange(60, dtype=np.uint8) * 4, (40, 1))                          # linear ramps stay linear
    out = oracle.resize_linear(ramp, 40, 50).astype(np.float64)
    assert np.abs(np.diff(out[5, 2:-2]) - 4 * 60 / 50).max() <= 1.0


def _fast_bruteforce(img, thr):
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = int(img[y, x])
            ring = [int(img[y + dy, x + dx]) for dx, dy in RING]
            best = 0
            for t in range(255, thr, -1):  # largest t for which 9 contiguous ring pixels are all brighter / darker
                br = [r > v + t - 1 for r in ring]
                dk = [r < v - t + 1 for r in ring]
                ok = any(all(b[(s + j) % 16] for j in range(9)) for b in (br, dk) for s in range(16))
                if ok:
                    best = t
                    break
            out[y, x] = best - 1 if best > thr else 0
    return out


def test_fast_score_against_bruteforce_definition():
    rng = np.random.default_rng(1)
    img = ndi.gaussian_filter(rng.random((24, 28)) * 255, 1.0).astype(np.uint8)
    img[8:14, 9:16] = 250
    img[15:19, 3:8] = 3
    got = oracle.fast_score_map(img, 20)
    assert np.array_equal(got, _fast_bruteforce(img, 20))
    assert got.max() > 50 and got[:3].max() == 0 and got[:, -3:].max() == 0


def test_detect_invariants_orientation_and_order():
    rng = np.random.default_rng(2)
    img = ndi.gaussian_filter(rng.random((122, 400)) * 255, 1.2)
    img = np.clip((img - img.mean()) * 6 + 128, 0, 255).astype(np.uint8)
    bits = np.zeros((122, 400), np.uint32)
    bits[:, :210] |= 1
    bits[:, 200:] |= 2
    bits[:, 100:110] = 0
    res = oracle.orb_detect(img, bits, 2, 120)
    quotas = oracle.orb_quotas(120)
    for m, (kp, resp) in enumerate(res):
        assert len(kp) > 30
        lv = kp[:, 3].astype(int)
        assert np.all(np.diff(lv) >= 0)                                   # ordered by level
        for l in np.unique(lv):
            sel = lv == l
            assert sel.sum() <= 2 * quotas[l]
            assert np.all(np.diff(resp[sel]) <= 0)                        # response descending inside a level
            s = 1.2 ** l
            xl, yl = kp[sel, 0] / s, kp[sel, 1] / s
            hl, wl = oracle.orb_level_size(122, 400, l)
            assert np.all((xl >= 31 - 1e-3) & (xl < wl - 31 + 1e-3) & (yl >= 31 - 1e-3) & (yl < hl - 31 + 1e-3))
        l0 = kp[lv == 0]
        assert np.all(((bits[l0[:, 1].astype(int), l0[:, 0].astype(int)] >> m) & 1) == 1)
        assert np.all((kp[:, 2] >= 0) & (kp[:, 2] <= 360))
    # orientation: intensity centroid of a horizontal / vertical ramp
    ramp_x = np.tile(np.arange(100, dtype=np.uint8) * 2, (100, 1))
    ramp_x[50, 50] = 255
    kx = oracle.orb_detect(ramp_x, np.ones((100, 100), np.uint32), 1, 50)[0][0]
    assert len(kx) >= 1 and np.all(np.minimum(kx[:, 2], 360 - kx[:, 2]) < 1.0)        # centroid along +x -> 0 deg
    ky = oracle.orb_detect(np.ascontiguousarray(ramp_x.T), np.ones((100, 100), np.uint32), 1, 50)[0][0]
    assert len(ky) >= 1 and np.all(np.abs(ky[:, 2] - 90) < 1.0)                        # along +y -> 90 deg


def test_describe_levels_matches_single_level_describe():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (122, 300), dtype=np.uint8)
    kp = np.array([[50, 40, 0.0, 0], [100.4, 60.6, 90.0, 0], [30.0, 60, 0, 0], [200, 45, 37.0, 1]], np.float32)
    desc, kept = oracle.orb_describe_levels(img, kp)
    assert kept.tolist() == [0, 1, 3]
    d0, _ = oracle.orb_describe(oracle.gauss7(img), kp[:1, :2], 1.0, 0.0)
    assert np.array_equal(desc[0], d0[0])
    c, s = np.float32(np.cos(np.float64(np.float32(90.0) * np.float32(np.pi / 180)))), np.float32(1.0)
    d1, _ = oracle.orb_describe(oracle.gauss7(img), kp[1:2, :2], c, s)
    assert np.array_equal(desc[1], d1[0])
    h1, w1 = oracle.orb_level_size(122, 300, 1)
    lvl1 = oracle.gauss7(oracle.resize_linear(img, h1, w1))
    a = np.float32(37.0) * np.float32(np.pi / 180)
    ca, sa = np.float32(np.cos(np.float64(a))), np.float32(np.sin(np.float64(a)))
    cx, cy = int(np.rint(np.float32(200) * np.float32(1 / np.float32(1.2)))), int(np.rint(np.float32(45) * np.float32(1 / np.float32(1.2))))
    d2, _ = oracle.orb_describe(lvl1, np.array([[cx, cy]], np.float32), ca, sa)
    assert np.array_equal(desc[2], d2[0])
