"""CPU: the host-side mirror of the reference's model classes against golden vectors captured from the
reference (tests/golden/geometry_c2.npz).  The unwrap LUT is built with the same numpy float32/float64
operations as the reference, so it is compared BIT-EXACT."""
import os

import numpy as np

from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
from vo_single_camera_sos_amd.omnistereo.panorama import Panorama

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry_c2.npz"))


def _rig():
    gs = synthetic_gums()
    gs.top_model.panorama = Panorama(gs.top_model, width=1200)
    gs.bot_model.panorama = Panorama(gs.bot_model, width=1200)
    return gs


def test_panorama_dimensions_and_lut_bit_exact():
    gs = _rig()
    for name, m in (("top", gs.top_model), ("bot", gs.bot_model)):
        pn = m.panorama
        rows, cols, px, hmax, hmin, circ = G[name + "_pano"]
        assert (pn.rows, pn.cols) == (int(rows), int(cols))
        assert pn.pixel_size == px and pn.cyl_height_max == hmax and pn.z_height_min == hmin
        assert pn.cyl_circumference == circ
        sel = G[name + "_lut_rows"]
        assert np.array_equal(pn.world2cam_LUT_map_x[sel], G[name + "_lut_x"], equal_nan=True)
        assert np.array_equal(pn.world2cam_LUT_map_y[sel], G[name + "_lut_y"], equal_nan=True)
        assert int(np.isnan(pn.world2cam_LUT_map_x).sum()) == int(G[name + "_lut_nan_count"][0])
        assert np.array_equal(m.T_model_wrt_C, G[name + "_T_model_wrt_C"])
        assert np.array_equal(m.F[:3, 0], G[name + "_F"])


def test_pixel_to_angles_bearings_and_projection():
    gs = _rig()
    for name, m in (("top", gs.top_model), ("bot", gs.bot_model)):
        mm = G["m_" + name]
        az, el = m.panorama.get_direction_angles_from_pixel_pano(mm, use_LUTs=False)
        assert np.array_equal(az, G["az_" + name], equal_nan=True)
        assert np.array_equal(el, G["el_" + name], equal_nan=True)
        b = m.get_3D_point_from_angles_wrt_focus(azimuth=az, elevation=el)[0]
        assert np.array_equal(b, G["bearing_" + name], equal_nan=True)
    Pw = np.array([[[2000.0, 500.0, 300.0, 1.0]]])
    u_t, v_t, _ = gs.top_model.get_pixel_from_3D_point_wrt_C(Pw)
    u_b, v_b, _ = gs.bot_model.get_pixel_from_3D_point_wrt_C(Pw)
    assert np.allclose([u_t.item(), v_t.item(), u_b.item(), v_b.item()], G["roundtrip_px"], rtol=1e-14)


def test_product_orb_pattern_equals_oracle_pattern():
    import oracle
    from vo_single_camera_sos_amd import orb_pattern
    assert np.array_equal(orb_pattern.orb_pattern(), oracle.orb_pattern())
    ca, sa = orb_pattern.angle_cos_sin(-1.0)
    assert ca.dtype == np.float32 and np.isclose(ca, np.cos(np.deg2rad(1.0))) and sa < 0


def test_annulus_masks_and_bucket_columns():
    gs = _rig()
    top, bot = gs.make_annulus_masks((480, 640))
    assert top.dtype == np.uint8 and set(np.unique(top)) == {0, 255}
    assert not np.any((top > 0) & (bot > 0))          # the two annuli are disjoint
    assert top[239, 319 + 200] == 255 and top[239, 319 + 100] == 0 and bot[239, 319 + 80] == 255
    pn = gs.top_model.panorama
    assert pn.get_panorama_col_from_azimuth(0.0) == 1199
    assert pn.get_panorama_col_from_azimuth(np.pi) == 599
    assert pn.get_panorama_col_from_azimuth(2 * np.pi - 1e-9) in (0, 1)
