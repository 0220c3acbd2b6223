"""Generates tests/golden/thirdparty_mineigen.npz: the minimum-eigenvalue corner response (K4a: what cv2.goodFeaturesToTrack
thresholds and sorts) on real photographs by scipy.ndimage in double precision, under /opt/conda/bin/python3.9 of the build
container:

    /opt/conda/bin/python3.9 tests/golden/make_thirdparty_mineigen.py

3 x 3 Sobel derivatives scaled by 1 / (2^(3-1) * 3 * 255), their products summed over the 3 x 3 block (not normalised), both
with the reflect-101 border; lambda_min = (a + c) - sqrt((a - c)^2 + b^2) with a = Sxx / 2, b = Sxy, c = Syy / 2.  The oracle (and the
HIP kernel, bit for bit with it) evaluates the same in float32 with a pinned operation order; the fixture holds the float64 values
at every pixel of the 1-px border ring (where the border rule decides) and at 3000 random interior pixels per photograph, plus
the map's maximum (what the quality level multiplies).  The grey crops are thirdparty_orientation.npz's."""
import os

import numpy as np
import scipy.ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    G = np.load(os.path.join(HERE, "thirdparty_orientation.npz"))
    rng = np.random.default_rng(4242)
    out = {}
    sx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=np.float64)
    scale = 1.0 / (4 * 3 * 255.0)
    box = np.ones((3, 3))
    for tag in ("camera", "astronaut", "coffee"):
        g = G[tag + "_gray"].astype(np.float64)
        dx, dy = ndi.correlate(g, sx, mode="mirror") * scale, ndi.correlate(g, sx.T, mode="mirror") * scale
        a = 0.5 * ndi.correlate(dx * dx, box, mode="mirror")
        b = ndi.correlate(dx * dy, box, mode="mirror")
        c = 0.5 * ndi.correlate(dy * dy, box, mode="mirror")
        eig = (a + c) - np.sqrt((a - c) ** 2 + b * b)
        rows, cols = g.shape
        ring = np.zeros(g.shape, bool)
        ring[0, :] = ring[-1, :] = ring[:, 0] = ring[:, -1] = True
        ys, xs = np.nonzero(ring)
        yi, xi = rng.integers(1, rows - 1, 3000), rng.integers(1, cols - 1, 3000)
        y, x = np.concatenate([ys, yi]), np.concatenate([xs, xi])
        out[tag + "_yx"] = np.stack([y, x], axis=1).astype(np.int16)
        out[tag + "_eig"] = eig[y, x]
        out[tag + "_max"] = np.array(eig.max())
        print(tag, g.shape, len(y), "samples, max", eig.max())
    p = os.path.join(HERE, "thirdparty_mineigen.npz")
    np.savez_compressed(p, **out)
    print(p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
