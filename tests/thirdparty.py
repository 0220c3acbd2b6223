"""Loaders and the third-party-derived expectations shared by tests/test_oracle_thirdparty.py (oracle, CPU) and
tests/test_gpu_thirdparty.py (HIP path through the C ABI).  The fixtures come from scikit-image / scipy on real photographs
(tests/golden/make_thirdparty_fixtures.py); the only arithmetic added here is the integer BGR -> gray formula the product
documents (include/sosvo.h, K3) and connected components (scipy) for the FAST set comparison."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, "thirdparty_%s.npz" % name))


def gray_of(bgr):
    """(1868 B + 9617 G + 4899 R + 8192) >> 14 on a [rows, cols, 3] array taken as B, G, R."""
    b, g, r = (bgr[..., c].astype(np.int64) for c in range(3))
    return ((1868 * b + 9617 * g + 4899 * r + 8192) >> 14).astype(np.uint8)


def median_cases():
    """-> list of (tag, ksize, input [rows, cols, 3] u8, expected gray [rows, cols] u8).  A colour photograph is handed over
    as it is (its channels taken as B, G, R: the median is per channel, so the naming does not matter) and the expectation
    is the gray formula applied to the third-party per-channel medians; the grey photograph is replicated into three
    channels, for which the formula is the identity (1868 + 9617 + 4899 = 16384) and the expectation IS the median."""
    M = load("median")
    out = []
    for tag in ("astronaut", "coffee", "camera"):
        img = M[tag + "_in"]
        for k in (11, 5, 3):
            med = M[tag + "_median%d" % k]
            if img.ndim == 2:
                out.append((tag, k, np.ascontiguousarray(np.repeat(img[..., None], 3, axis=2)), med))
            else:
                out.append((tag, k, img, gray_of(med)))
    return out


def fast_cases():
    """-> list of (tag, threshold, gray u8, corner set bool [rows, cols]) from skimage.feature.corner_fast(n=9)."""
    F = load("fast")
    out = []
    for tag in ("camera", "astronaut", "coffee"):
        g = F[tag + "_gray"]
        for t in (10, 20, 40):
            c = np.unpackbits(F[tag + "_corners_t%d" % t], axis=1)[:, : g.shape[1]].astype(bool)
            assert int(c.sum()) == int(F[tag + "_count_t%d" % t][0])
            out.append((tag, t, g, c))
    return out


def check_fast_keypoints(kp_xy, corners, border=3, score=None):
    """FAST-9/16 + 3x3 non-maximum suppression against a third-party corner SET: (i) every keypoint is a corner of the set;
    (ii) every 8-connected component of the set (inside the border FAST cannot evaluate) holds a keypoint, except
    components whose maximum score is a TIE between touching pixels (OpenCV's strict-maximum rule keeps neither; common at
    low thresholds, where scores are small integers).  With `score` (a corner-score map) every empty component is checked
    to be such a tie; without it at most 8 % of the components may be empty.  -> (keypoints, components, empty)."""
    import scipy.ndimage as ndi
    rows, cols = corners.shape
    inner = np.zeros_like(corners)
    inner[border:rows - border, border:cols - border] = corners[border:rows - border, border:cols - border]
    xs, ys = kp_xy[:, 0].astype(int), kp_xy[:, 1].astype(int)
    assert np.array_equal(kp_xy, np.stack([xs, ys], axis=1).astype(np.float32))          # integer pixel positions
    assert inner[ys, xs].all(), "a keypoint that is not a FAST-9 corner of the third-party set"
    lab, ncomp = ndi.label(inner, structure=np.ones((3, 3), dtype=int))
    hit = np.zeros(ncomp + 1, dtype=bool)
    hit[lab[ys, xs]] = True
    empty = np.flatnonzero(~hit[1:]) + 1
    if score is None:
        assert len(empty) <= max(1, (8 * ncomp) // 100), (len(empty), ncomp)
    else:
        for c in empty:
            cy, cx = np.nonzero(lab == c)
            top = score[cy, cx] == score[cy, cx].max()
            ty, tx = cy[top], cx[top]
            assert len(ty) >= 2, "a component without a keypoint whose maximum is unique"
            d = np.maximum(np.abs(ty[:, None] - ty[None, :]), np.abs(tx[:, None] - tx[None, :]))
            assert ((d == 1).sum(axis=1) >= 1).all(), "maxima of an empty component that do not touch"
    return len(xs), ncomp, len(empty)


def hamming_cases():
    """-> list of (tag, q [nq, 32] u8, t [nt, 32] u8, nn2_idx [nq, 2], nn2_dist [nq, 2]) from skimage ORB descriptors,
    skimage.feature.match_descriptors and scipy's cdist (first index wins ties: numpy's stable argsort / argmin)."""
    Hm = load("hamming")
    out = []
    for tag in ("motorcycle", "astronaut"):
        assert np.array_equal(Hm[tag + "_match_t"], Hm[tag + "_nn2_idx"][:, 0])
        out.append((tag, Hm[tag + "_q"], Hm[tag + "_t"], Hm[tag + "_nn2_idx"], Hm[tag + "_nn2_dist"]))
    return out


def keys_of(idx, dist, shift=20):
    return ((dist.astype(np.uint32) << shift) | idx.astype(np.uint32)).astype(np.uint32)


def orientation_cases():
    """-> list of (tag, gray u8, xy [n, 2] int32 level-0 ORB keypoints, angle_deg [n] float64) from
    skimage.feature.corner_orientations with OFAST_MASK (tests/golden/make_thirdparty_orientation.py)."""
    O = load("orientation")
    return [(tag, O[tag + "_gray"], O[tag + "_xy"], O[tag + "_angle_deg"]) for tag in ("camera", "astronaut", "coffee")]


def harris_cases():
    """-> list of (tag, gray u8, xy [n, 2], harris [n] float64): ORB's 7 x 7 Harris response at the level-0 keypoints by
    scipy.ndimage (Sobel pair, box sums, float64)."""
    O = load("orientation")
    return [(tag, O[tag + "_gray"], O[tag + "_xy"], O[tag + "_harris"]) for tag in ("camera", "astronaut", "coffee")]


def gauss7_cases():
    """-> list of (tag, gray u8, blurred u8): the 7 x 7 fixed-point Gaussian (taps 18 34 49 54 49 34 18 / 256 per axis,
    reflect-101 border, one rounding) by scipy.ndimage.correlate1d on integers."""
    O = load("orientation")
    return [(tag, O[tag + "_gray"], O[tag + "_gauss7"]) for tag in ("camera", "astronaut", "coffee")]


def check_harris(kp4, resp, xy, harris, rtol=2e-6):
    l0 = kp4[:, 3] == 0
    assert int(l0.sum()) == xy.shape[0] and np.array_equal(kp4[l0][:, :2].astype(np.int32), xy)
    r = resp[l0].astype(np.float64)
    rel = np.abs(r - harris) / np.abs(harris)
    assert rel.max() <= rtol, float(rel.max())
    return float(rel.max())


def check_orientations(kp4, xy, angle_deg, tol_deg=0.02):
    """kp4 [n, 4] (x, y, angle, level) of a detector run on the fixture's image: its level-0 keypoints are the fixture's, in
    order, and their angles agree with the third-party atan2(m01, m10) within `tol_deg` on the circle (cv2's fastAtan2, which
    the product restates, is a degree-7 polynomial; measured on these photographs: 0.009 degrees at worst).  -> largest difference in degrees."""
    l0 = kp4[kp4[:, 3] == 0]
    assert l0.shape[0] == xy.shape[0] and np.array_equal(l0[:, :2].astype(np.int32), xy)
    d = np.abs(l0[:, 2].astype(np.float64) - angle_deg)
    d = np.minimum(d, 360.0 - d)
    assert d.max() <= tol_deg, (float(d.max()), int(d.argmax()))
    return float(d.max())


def brief_cases():
    """-> list of (tag, gray [rows, cols] u8, xy [n, 2] i32, angles_deg list, desc per angle [n, 32] u8, angle_each_deg [n] f32,
    desc_each [n, 32] u8): rotated-BRIEF descriptors by scikit-image's descriptor loop on the grey crops of the orientation
    fixture after the 7 x 7 fixed-point Gaussian (tests/golden/make_thirdparty_brief.py)."""
    B, G = load("brief"), load("orientation")
    angles = [float(a) for a in B["angles_deg"]]
    return [(tag, G[tag + "_gray"], B[tag + "_xy"], angles, [B[tag + "_desc_%d" % k] for k in range(len(angles))],
             B[tag + "_angle_each_deg"], B[tag + "_desc_each"]) for tag in ("camera", "astronaut", "coffee")]


def mineigen_cases():
    """-> list of (tag, gray [rows, cols] u8, yx [n, 2] i16, eig [n] f64, max f64): the minimum-eigenvalue corner response by
    scipy.ndimage in double precision at the border ring and 3000 interior pixels (tests/golden/make_thirdparty_mineigen.py)."""
    M, G = load("mineigen"), load("orientation")
    return [(tag, G[tag + "_gray"], M[tag + "_yx"], M[tag + "_eig"], float(M[tag + "_max"])) for tag in ("camera", "astronaut", "coffee")]
