"""Generates tests/golden/thirdparty_*.npz with INDEPENDENT third-party implementations on real photographs: scikit-image
0.18.3 + scipy 1.7.1 under /opt/conda/bin/python3.9 of the build container (neither is importable by the interpreter the
tests run on, and nothing of them travels: the fixtures hold inputs and expected outputs only).

    /opt/conda/bin/python3.9 tests/golden/make_thirdparty_fixtures.py

What they pin: the HIP path's byte / index arithmetic against code that shares
nothing with this repository's oracle.  None of it is an OpenCV binary -- K1-K10 stay "parity unpinned" against those --
but the semantics compared are the ones OpenCV documents:
  median     skimage.filters.median (scipy.ndimage.median_filter), 11 x 11 and 5 x 5 window, mode 'nearest' = replicated
             border, per channel, on colour and grey photographs                -> cv2.medianBlur (camera_models.py:1711)
  fast       skimage.feature.corner_fast(n=9) corner SETS at thresholds 10 / 20 / 40 on integer-valued float images (the
             comparisons are then exact: brighter iff I_p > I_c + t)            -> FAST-9/16 (camera_models.py:1664)
  hamming    skimage.feature.ORB descriptors of two real image pairs, skimage.feature.match_descriptors(metric='hamming',
             cross_check=False) and scipy.spatial.distance.cdist for 1-NN / 2-NN  -> BFMatcher(NORM_HAMMING) (:402-446)
  warp       scipy.ndimage.map_coordinates(order=1, mode='grid-constant', cval=0) and skimage.transform.warp at positions on
             the exact 1/32-pixel grid, where OpenCV's fixed-point remap IS the rounded exact bilinear value -> cv2.remap (:293)
"""
import os

import numpy as np
import scipy.ndimage as ndi
import skimage
import skimage.data
from scipy.spatial.distance import cdist
from skimage.feature import ORB, corner_fast, match_descriptors
from skimage.filters import median as sk_median
from skimage.transform import warp

HERE = os.path.dirname(os.path.abspath(__file__))


def photo(name):
    img = skimage.io.imread(os.path.join(os.path.dirname(skimage.data.__file__), name))
    return np.ascontiguousarray(img)


def grey_u8(rgb):
    """Integer luma of an RGB photograph (only to obtain a grey INPUT; not under test)."""
    return ((rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 2].astype(np.int64) * 1868 + 8192) >> 14).astype(np.uint8)


def main():
    import skimage.io  # noqa: F401
    versions = np.array(["scikit-image " + skimage.__version__, "scipy " + __import__("scipy").__version__, "numpy " + np.__version__])
    astro = photo("astronaut.png")[..., :3]            # RGB 512 x 512
    coffee = photo("coffee.png")[..., :3]              # RGB 400 x 600
    camera = photo("camera.png")                       # grey 512 x 512
    moto_l, moto_r = photo("motorcycle_left.png")[..., :3], photo("motorcycle_right.png")[..., :3]

    # ---- median (per channel, replicated border)
    out = {"versions": versions}
    for tag, img in (("astronaut", astro[40:168, 150:350]), ("coffee", coffee[100:190, 200:331]), ("camera", camera[90:219, 170:333])):
        img = np.ascontiguousarray(img)
        out[tag + "_in"] = img
        for k in (11, 5, 3):
            if img.ndim == 3:
                res = np.stack([sk_median(img[..., c], selem=np.ones((k, k), dtype=np.uint8), mode="nearest") for c in range(3)], axis=-1)
            else:
                res = sk_median(img, selem=np.ones((k, k), dtype=np.uint8), mode="nearest")
            assert res.dtype == np.uint8
            out[tag + "_median%d" % k] = res
    np.savez_compressed(os.path.join(HERE, "thirdparty_median.npz"), **out)

    # ---- FAST-9 corner sets
    out = {"versions": versions}
    for tag, g in (("camera", camera[100:260, 150:390]), ("astronaut", grey_u8(astro)[0:200, 120:400]), ("coffee", grey_u8(coffee)[60:240, 150:450])):
        g = np.ascontiguousarray(g)
        out[tag + "_gray"] = g
        for t in (10, 20, 40):
            resp = corner_fast(g.astype(np.float64), n=9, threshold=float(t))   # float image with integer values: exact compares
            out[tag + "_corners_t%d" % t] = np.packbits(resp > 0, axis=1)
            out[tag + "_count_t%d" % t] = np.array([int((resp > 0).sum())])
    np.savez_compressed(os.path.join(HERE, "thirdparty_fast.npz"), **out)

    # ---- Hamming nearest neighbours on real ORB descriptors
    out = {"versions": versions}
    pairs = (("motorcycle", grey_u8(moto_l)[100:400, 100:600], grey_u8(moto_r)[100:400, 100:600], 700),
             ("astronaut", grey_u8(astro), np.ascontiguousarray(grey_u8(astro)[8:500, 5:490]), 450))
    for tag, a, b, nk in pairs:
        ds = []
        for im in (a, b):
            orb = ORB(n_keypoints=nk, fast_threshold=0.05)
            orb.detect_and_extract(im.astype(np.float64) / 255.0)
            ds.append(orb.descriptors.astype(bool))
        q, t = ds
        D = np.rint(cdist(q, t, metric="hamming") * 256).astype(np.int64)             # [nq, nt] exact integers
        m = match_descriptors(q, t, metric="hamming", cross_check=False)                # [nq, 2] (query, train): first minimum
        order = np.argsort(D, axis=1, kind="stable")[:, :2]
        out[tag + "_q"] = np.packbits(q, axis=1, bitorder="little")                     # bit k of byte j = test 8 j + k
        out[tag + "_t"] = np.packbits(t, axis=1, bitorder="little")
        out[tag + "_match_q"], out[tag + "_match_t"] = m[:, 0].astype(np.int32), m[:, 1].astype(np.int32)
        out[tag + "_nn2_idx"] = order.astype(np.int32)
        out[tag + "_nn2_dist"] = np.take_along_axis(D, order, axis=1).astype(np.int32)
        assert np.array_equal(m[:, 0], np.arange(q.shape[0])) and np.array_equal(m[:, 1], order[:, 0])
    np.savez_compressed(os.path.join(HERE, "thirdparty_hamming.npz"), **out)

    # ---- bilinear sampling on the 1/32-pixel grid
    out = {"versions": versions}
    rng = np.random.default_rng(20261004)
    img = np.ascontiguousarray(coffee[120:216, 210:338])                               # 96 x 128 x 3
    H, W = img.shape[:2]
    rows, cols = 60, 90
    mx = (rng.integers(-2 * 32, (W + 1) * 32, (rows, cols)) / 32.0).astype(np.float32)   # some taps outside the image
    my = (rng.integers(-2 * 32, (H + 1) * 32, (rows, cols)) / 32.0).astype(np.float32)
    mx[0, :8] = [0.0, W - 1.0, W - 1.0 + 1 / 32.0, -1 / 32.0, W - 0.5, -0.5, W, -1.0]
    my[0, :8] = [0.0, H - 1.0, 3.0, 5.25, H - 0.5, -0.5, 2.0, 4.0]
    exact = np.stack([ndi.map_coordinates(img[..., c].astype(np.float64), [my.astype(np.float64), mx.astype(np.float64)], order=1,
                                          mode="grid-constant", cval=0.0, prefilter=False) for c in range(3)], axis=-1)
    out["img"], out["map_x"], out["map_y"] = img, mx, my
    out["bilinear_exact"] = exact                                                         # float64, exact dyadic values
    out["bilinear_rounded"] = np.floor(exact + 0.5).astype(np.uint8)
    inside = (mx >= 0) & (mx <= W - 1) & (my >= 0) & (my <= H - 1)
    sk = np.stack([warp(img[..., c].astype(np.float64), np.stack([my, mx]).astype(np.float64), order=1, mode="constant", cval=0.0,
                        preserve_range=True) for c in range(3)], axis=-1)
    assert np.array_equal(sk[inside], exact[inside])                                      # skimage.transform.warp agrees where both are defined
    out["inside"] = inside
    np.savez_compressed(os.path.join(HERE, "thirdparty_warp.npz"), **out)
    for f in ("median", "fast", "hamming", "warp"):
        p = os.path.join(HERE, "thirdparty_%s.npz" % f)
        print(p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
