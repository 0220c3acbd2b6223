"""Generates tests/golden/thirdparty_orientation.npz: intensity-centroid ORIENTATIONS of ORB keypoints on real photographs by
scikit-image's `corner_orientations` with its `OFAST_MASK` (the radius-15 disc of the oFAST paper -- row half-widths 3, 6, 8,
9, 10, 11, 12, 13, 13, 14, 14, 14, 15, 15, 15, 15, the table OpenCV's ORB uses), under /opt/conda/bin/python3.9 of the build
container:

    /opt/conda/bin/python3.9 tests/golden/make_thirdparty_orientation.py

The keypoint POSITIONS come from this repository's CPU oracle (level-0 keypoints of oracle.orb_detect: the GPU detector
returns the same ones bit for bit, tests/test_gpu_orb.py); what the fixture pins is the angle at those positions:
atan2(m01, m10) over the disc by independent code, in double precision.  cv2's ORB evaluates the same moments in integers
and the angle with fastAtan2 (a degree-7 polynomial, documented accuracy ~0.3 degrees), which is what oracle / HIP path
restate: the tests compare within 0.02 degrees on the circle.  Nothing of scikit-image travels: inputs and angles only."""
import os
import sys

import numpy as np
import scipy.ndimage as ndi
import skimage
import skimage.data
import skimage.io
from skimage.feature import corner_orientations
from skimage.feature.orb import OFAST_MASK

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402  (ctypes face of oracle/*.c: positions only)


def photo(name):
    return np.ascontiguousarray(skimage.io.imread(os.path.join(os.path.dirname(skimage.data.__file__), name)))


def grey_u8(rgb):
    return ((rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 2].astype(np.int64) * 1868 + 8192) >> 14).astype(np.uint8)


def main():
    out = {"versions": np.array(["scikit-image " + skimage.__version__, "numpy " + np.__version__])}
    cases = (("camera", photo("camera.png")[60:380, 100:500]), ("astronaut", grey_u8(photo("astronaut.png")[..., :3])[0:300, 80:480]),
             ("coffee", grey_u8(photo("coffee.png")[..., :3])[40:340, 100:560]))
    for tag, g in cases:
        g = np.ascontiguousarray(g)
        bits = np.ones(g.shape, dtype=np.uint32)
        kp4, resp = oracle.orb_detect(g, bits, 1, 500, 1024)[0]
        lvl0 = kp4[:, 3] == 0
        xy = kp4[lvl0][:, :2].astype(np.int64)                      # level 0: integer pixel positions
        corners = np.stack([xy[:, 1], xy[:, 0]], axis=1)            # (row, col)
        ang = corner_orientations(g.astype(np.float64), corners, OFAST_MASK)   # radians, atan2(m01, m10)
        out[tag + "_gray"] = g
        out[tag + "_xy"] = xy.astype(np.int32)
        out[tag + "_angle_deg"] = np.mod(np.degrees(ang), 360.0)
        # Harris response of the 7 x 7 block around each keypoint, k = 0.04, gradients by the 3 x 3 Sobel pair, scaled by
        # (1 / (4 * 7 * 255))^4 -- ORB's HarrisResponses -- with scipy's correlate and plain float64 sums
        gi = g.astype(np.float64)
        sx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=np.float64)
        ix, iy = ndi.correlate(gi, sx, mode="mirror"), ndi.correlate(gi, sx.T, mode="mirror")
        box = np.ones((7, 7))
        a, b, c = (ndi.correlate(v, box, mode="constant") for v in (ix * ix, iy * iy, ix * iy))
        sc4 = (1.0 / (4 * 7 * 255.0)) ** 4
        r = ((a * b - c * c) - 0.04 * (a + b) ** 2) * sc4
        out[tag + "_harris"] = r[xy[:, 1], xy[:, 0]]
        # 7 x 7 Gaussian in 8.8 fixed point (taps 18 34 49 54 49 34 18 / 256 per axis, reflect-101 border, one rounding at the
        # end): scipy's correlate1d on integers with mode 'mirror'
        taps = np.array([18, 34, 49, 54, 49, 34, 18], dtype=np.int64)
        v = ndi.correlate1d(ndi.correlate1d(g.astype(np.int64), taps, axis=1, mode="mirror"), taps, axis=0, mode="mirror")
        out[tag + "_gauss7"] = ((v + 32768) >> 16).astype(np.uint8)
        print(tag, g.shape, "level-0 keypoints", len(xy))
    p = os.path.join(HERE, "thirdparty_orientation.npz")
    np.savez_compressed(p, **out)
    print(p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
