"""Generates tests/golden/thirdparty_brief.npz: rotated-BRIEF DESCRIPTORS of keypoints on real photographs by scikit-image's
descriptor loop (`skimage.feature.orb_cy._orb_loop`, the Cython core of `skimage.feature.ORB.extract`: its test table is
`orb_descriptor_positions.txt`, OpenCV's learned `bit_pattern_31_`), under /opt/conda/bin/python3.9 of the build container:

    /opt/conda/bin/python3.9 tests/golden/make_thirdparty_brief.py

What the fixture pins is K6's sampling rule: the table and which of its columns is x and which y, the rotation of a test
point by the keypoint's angle (x' = x cos - y sin, y' = x sin + y cos, rounded to the nearest pixel), the comparison
(I(p0) < I(p1) sets the bit) and the order of the 256 bits in the 32 bytes.  The sampled image is the photograph's grey crop
after the 7 x 7 fixed-point Gaussian computed HERE by scipy on integers (the filter itself is pinned by
thirdparty_orientation.npz's `*_gauss7` arrays); positions are random interior pixels, angles a fixed list plus one random
angle per keypoint.  scikit-image evaluates the rotation in double precision, cv2 (and oracle / HIP path) in single: a rotated
coordinate within ~1e-6 of a half-integer could round apart -- none of the 1.8 M test points of this fixture does (the
tests demand equality).  Nothing of scikit-image travels: inputs and descriptor bytes only."""
import os

import numpy as np
import scipy.ndimage as ndi
import skimage
import skimage.data
import skimage.io
from skimage.feature.orb_cy import _orb_loop

HERE = os.path.dirname(os.path.abspath(__file__))
ANGLES_DEG = (0.0, -1.0, 17.3, 45.0, 90.0, 123.456, 200.0, 271.25)   # (-1: the angle cv2 gives a GFT keypoint)


def photo(name):
    return np.ascontiguousarray(skimage.io.imread(os.path.join(os.path.dirname(skimage.data.__file__), name)))


def grey_u8(rgb):
    return ((rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 2].astype(np.int64) * 1868 + 8192) >> 14).astype(np.uint8)


def gauss7(g):
    taps = np.array([18, 34, 49, 54, 49, 34, 18], dtype=np.int64)
    v = ndi.correlate1d(ndi.correlate1d(g.astype(np.int64), taps, axis=1, mode="mirror"), taps, axis=0, mode="mirror")
    return ((v + 32768) >> 16).astype(np.uint8)


def describe(blurred, xy, angles_rad):
    kp = np.ascontiguousarray(np.stack([xy[:, 1], xy[:, 0]], axis=1).astype(np.intp))        # (row, col)
    bits = _orb_loop(np.ascontiguousarray(blurred.astype(np.float64)), kp, np.ascontiguousarray(angles_rad, dtype=np.float64))
    return np.packbits(np.asarray(bits).astype(np.uint8), axis=1, bitorder="little")           # test 8 i + j -> bit j of byte i


def main():
    out = {"versions": np.array(["scikit-image " + skimage.__version__, "numpy " + np.__version__]),
           "angles_deg": np.array(ANGLES_DEG, dtype=np.float64)}
    rng = np.random.default_rng(20261004)
    cases = (("camera", photo("camera.png")[60:380, 100:500]), ("astronaut", grey_u8(photo("astronaut.png")[..., :3])[0:300, 80:480]),
             ("coffee", grey_u8(photo("coffee.png")[..., :3])[40:340, 100:560]))
    same = np.load(os.path.join(HERE, "thirdparty_orientation.npz"))
    for tag, g in cases:
        g = np.ascontiguousarray(g)
        assert np.array_equal(g, same[tag + "_gray"])
        bl = gauss7(g)
        n = 256
        xy = np.stack([rng.integers(31, g.shape[1] - 31, n), rng.integers(31, g.shape[0] - 31, n)], axis=1).astype(np.int32)
        out[tag + "_xy"] = xy                  # (the grey crops are thirdparty_orientation.npz's `*_gray`: the same slices)
        # the angle in radians as cv2 forms it: float32 degrees times float32 (pi / 180)
        for k, deg in enumerate(ANGLES_DEG):
            a = float(np.float32(deg) * np.float32(np.pi / 180.0))
            out[tag + "_desc_%d" % k] = describe(bl, xy, np.full(n, a))
        deg_each = rng.uniform(0.0, 360.0, n).astype(np.float32)
        out[tag + "_angle_each_deg"] = deg_each
        out[tag + "_desc_each"] = describe(bl, xy, (deg_each * np.float32(np.pi / 180.0)).astype(np.float64))
        print(tag, g.shape, n, "keypoints x", len(ANGLES_DEG) + 1, "angle sets")
    p = os.path.join(HERE, "thirdparty_brief.npz")
    np.savez_compressed(p, **out)
    print(p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
