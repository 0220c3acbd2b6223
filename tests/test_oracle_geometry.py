"""CPU: pin the oracle's geometry and score against golden vectors captured from the reference's
own numpy code (tests/golden/geometry_c2.npz, generator tests/golden/make_fixtures.py) and against
the known-answer values recorded in SURVEY.md Appendix D."""
import os

import numpy as np
import pytest

import oracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry_c2.npz"))
RT = dict(rtol=1e-12, atol=1e-12)


def _pano(name):
    rows, cols, px, hmax, hmin, _ = G[name + "_pano"]
    return int(cols), int(rows), px, hmax, hmin


def test_appendix_d_known_answers():
    cols, rows, px, hmax, hmin = _pano("top")
    assert (cols, rows) == (1200, 122) and px == 0.005235987755982988
    assert hmax == 0.2681943150929834 and hmin == -0.3683258284050759
    az, el = oracle.pano_to_angles([100.25, 700.0, 1199.5], [30.5, 80.0, 121.0], cols, rows, px, hmax)
    assert np.allclose(az, [5.7582775346422919, 2.6179938779914944, 2.6179938779922196e-03], rtol=1e-13)
    assert np.allclose(el, [0.10807394627744413, -0.14955951877280754, -0.3502926753423096], rtol=1e-13)
    b = oracle.angles_to_bearing(az, el)
    assert np.allclose(b[0], (0.8603213284306165, -0.49820943104826676, 0.10786368555701002), rtol=1e-13)
    az2, el2 = oracle.pano_to_angles([100.75, 699.0, 1198.0], [27.5, 70.0, 110.0], cols, rows, px, hmax)
    X = oracle.triangulate_midpoint(az, el, az2, el2, G["top_F"], G["bot_F"])
    assert np.allclose(X[0], (5354.191071306268, -3109.9618818261797, 820.42980176203434), rtol=1e-10)
    assert np.allclose(X[1], (-1639.8228690082908, 941.03692504020842, -135.39775591418149), rtol=1e-10)
    assert np.allclose(X[2], (1700.9728190964722, 11.131796091027001, -472.50877677119172), rtol=1e-10)
    assert float(G["thr_5deg"][0]) == 0.003805301908254455
    assert int(G["ransac_iters_3_065"][0]) == 210


def test_pano_pixel_to_angles_and_bearings():
    for name in ("top", "bot"):
        cols, rows, px, hmax, _ = _pano(name)
        m = G["m_" + name]
        az, el = oracle.pano_to_angles(m[:, 0], m[:, 1], cols, rows, px, hmax)
        assert np.array_equal(np.isnan(az), np.isnan(G["az_" + name]))
        assert np.array_equal(np.isnan(el), np.isnan(G["el_" + name]))
        assert np.allclose(az, G["az_" + name], equal_nan=True, **RT)
        assert np.allclose(el, G["el_" + name], equal_nan=True, **RT)
        b = oracle.angles_to_bearing(az, el)
        want = G["bearing_" + name]
        assert np.allclose(b, want[:, :3], equal_nan=True, **RT) and np.all(want[:, 3] == 1.0)


def test_triangulation_range_filter_and_gates():
    X = oracle.triangulate_midpoint(G["az_top"], G["el_top"], G["az_bot"], G["el_bot"], G["top_F"], G["bot_F"])
    want = G["tri_X_homo"]
    assert np.array_equal(np.isnan(X).any(1), np.isnan(want[:, :3]).any(1))
    assert np.allclose(X, want[:, :3], equal_nan=True, rtol=1e-9, atol=1e-7)
    assert np.array_equal(oracle.range_filter_homo(want[:, :3], 500.0, 7000.0), G["range_ok_500_7000"])
    assert np.array_equal(oracle.range_filter_homo(X, 500.0, 7000.0), G["range_ok_500_7000"])
    assert np.array_equal(oracle.range_filter_homo(X, 900.0, 0.0), G["range_ok_min_only"])
    mt, mb = G["m_top"][:, :2], G["m_bot"][:, :2]
    assert np.array_equal(oracle.pixel_gates(mt, mb, 1, 2.5), G["gate_stereo"])
    assert np.array_equal(oracle.pixel_gates(mt, mb, -1, 1.25), G["gate_f2f"])


def test_unwrap_lut_matches_reference_projection():
    for name in ("top", "bot"):
        cols, rows, px, hmax, hmin = _pano(name)
        lo, hi = G[name + "_elev"][:2]
        mx, my = oracle.unwrap_lut(G[name + "_params"], cols, rows, px, hmax, hmin, lo, hi)
        sel = G[name + "_lut_rows"]
        assert np.array_equal(np.isnan(mx[sel]), np.isnan(G[name + "_lut_x"]))
        # the reference evaluates the sphere point in float32 with numpy's float32 sin/cos; libm's
        # sinf/cosf may differ in the last float32 bit -> 1e-4 px (the product's host mirror builds the
        # LUT with the same numpy float32 ops and is checked bit-exact in test_host_mirror.py)
        assert np.allclose(mx[sel], G[name + "_lut_x"], equal_nan=True, rtol=0, atol=1e-4)
        assert np.allclose(my[sel], G[name + "_lut_y"], equal_nan=True, rtol=0, atol=1e-4)
        assert int(np.isnan(mx).sum()) == int(G[name + "_lut_nan_count"][0])
        assert np.isclose(np.nansum(mx.astype(np.float32), dtype=np.float64), G[name + "_lut_x_f32_sum"][0], rtol=1e-7)
    # Appendix D spot values
    cols, rows, px, hmax, hmin = _pano("top")
    lo, hi = G["top_elev"][:2]
    mx, my = oracle.unwrap_lut(G["top_params"], cols, rows, px, hmax, hmin, lo, hi)
    assert np.allclose((mx[10, 100], my[10, 100]), (503.26917562874735, 132.11399999410747), rtol=0, atol=1e-4)
    assert np.allclose((mx[121, 1199], my[121, 1199]), (433.0844411857262, 239.5), rtol=0, atol=1e-4)


@pytest.mark.parametrize("tag,is_z", [("z", True), ("radial", False)])
def test_rgbd_backprojection(tag, is_z):
    xyz, b = oracle.rgbd_backproject(G["rgbd_depth"], G["rgbd_u"], G["rgbd_v"], G["rgbd_intrinsics"], is_z)
    want = G["rgbd_xyz_" + tag][0]
    assert np.array_equal(np.isnan(xyz), np.isnan(want))
    assert np.allclose(xyz, want, equal_nan=True, rtol=1e-12, atol=1e-12)
    good = ~np.isnan(want[:, 2])
    assert np.allclose(b[good], G["rgbd_bearing_" + tag], rtol=1e-12, atol=1e-12)


def test_score_matches_reference_restatement():
    got = oracle.score_points(G["score_f"], G["score_p"], G["score_T"])
    assert np.allclose(got, G["score_expected"], rtol=0, atol=5e-16)


def test_trig_core_against_the_host_math_library():
    """oracle/trig_core.h (sin / cos / atan from + - * / only, shared with the device): within one ulp of numpy over the
    arguments of the path (azimuth in [0, 2 pi], elevation within +-pi/2, tan(elevation) a few units), exact at the
    awkward points."""
    import ctypes
    L = oracle.lib()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-7.0, 7.0, 200000), np.linspace(0, 2 * np.pi, 1441), [np.pi, 2 * np.pi, np.pi / 2, 0.0, -0.0]])
    s, c, a = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    L.orc_trig_eval(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(x.size), s.ctypes.data_as(ctypes.c_void_p),
                    c.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p))
    for got, want in ((s, np.sin(x)), (c, np.cos(x)), (a, np.arctan(x))):
        assert np.all(np.abs(got - want) <= np.spacing(np.abs(want))), np.abs(got - want).max()
    assert s[-5] == np.sin(np.pi) and s[-4] == np.sin(2 * np.pi) and c[-3] == np.cos(np.pi / 2)


def test_unwrap_lut_and_geometry_on_the_distorted_rig():
    """tests/golden/geometry_distorted.npz (reference-generated): k1..k3, xi1, xi2, alpha_c non-zero, gamma1 != gamma2,
    off-centre principal point -- the branch of orc_unwrap_lut the first fixture leaves at zero."""
    D = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry_distorted.npz"))
    for name in ("top", "bot"):
        rows, cols, px, hmax, hmin, _ = D[name + "_pano"]
        lo, hi = D[name + "_elev"][:2]
        mx, my = oracle.unwrap_lut(D[name + "_params"], int(cols), int(rows), px, hmax, hmin, lo, hi)
        sel = D[name + "_lut_rows"]
        assert np.array_equal(np.isnan(mx[sel]), np.isnan(D[name + "_lut_x"]))
        assert np.allclose(mx[sel], D[name + "_lut_x"], equal_nan=True, rtol=0, atol=1e-4)   # (float32 sinf / cosf, as above)
        assert np.allclose(my[sel], D[name + "_lut_y"], equal_nan=True, rtol=0, atol=1e-4)
        assert int(np.isnan(mx).sum()) == int(D[name + "_lut_nan_count"][0])
        assert np.array_equal(np.flatnonzero(np.isnan(mx).any(axis=1)), D[name + "_lut_nan_rows"])
        for tag, arr in (("x", mx), ("y", my)):
            assert np.isclose(np.nansum(arr.astype(np.float32), dtype=np.float64), D[name + "_lut_%s_f32_sum" % tag][0], rtol=1e-7)
        pano = (int(cols), int(rows), px, hmax)
        az, el = oracle.pano_to_angles(D["m_" + name][:, 0], D["m_" + name][:, 1], *pano)
        assert np.array_equal(np.isnan(az), np.isnan(D["az_" + name])) and np.array_equal(np.isnan(el), np.isnan(D["el_" + name]))
        assert np.allclose(az, D["az_" + name], equal_nan=True, **RT) and np.allclose(el, D["el_" + name], equal_nan=True, **RT)
        assert np.allclose(oracle.angles_to_bearing(az, el), D["bearing_" + name][:, :3], equal_nan=True, **RT)
    X = oracle.triangulate_midpoint(D["az_top"], D["el_top"], D["az_bot"], D["el_bot"], D["top_F"], D["bot_F"])
    want = D["tri_X_homo"]
    assert np.array_equal(np.isnan(X).any(1), np.isnan(want[:, :3]).any(1))
    assert np.allclose(X, want[:, :3], equal_nan=True, rtol=1e-9, atol=1e-7)
    assert np.array_equal(oracle.range_filter_homo(X, 500.0, 7000.0), D["range_ok_500_7000"])
    assert np.array_equal(oracle.pixel_gates(D["m_top"][:, :2], D["m_bot"][:, :2], 1, 2.5), D["gate_stereo"])
