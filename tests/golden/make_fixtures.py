"""Generates tests/golden/geometry_c2.npz by IMPORTING the reference's numpy geometry.

Runs ONLY in the build container (needs /root/reference); the GPU box never sees the reference.
    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py

cv2 / pyopengv / vispy are absent here, so they get placeholder modules: only the reference's
numpy-only methods are exercised (anything OpenCV/OpenGV-backed would return a mock and is not
recorded).  The fixture holds inputs and expected outputs only -- no reference source text.

Pins: a3 (unwrap LUT), a7 (pano pixel -> angles), a8 (angles -> bearing),
a9 (midpoint triangulation), a10 (range filter on homogeneous rows), a6's pixel gates, a12 (RGB-D
back-projection), a16 (RANSAC iteration budget / threshold), a17 (score definition).
"""
import collections
import collections.abc
import os
import sys
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "geometry_c2.npz")


def import_reference():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cv2 = mock.MagicMock(name="cv2")
    cv2.convertMaps.return_value = (None, None)
    cv2.INTER_LINEAR = 1
    cv2.BORDER_CONSTANT = 0
    sys.modules["cv2"] = cv2
    sys.modules["pyopengv"] = mock.MagicMock(name="pyopengv")
    vispy = types.ModuleType("vispy")
    scene = types.ModuleType("vispy.scene")

    class SceneCanvas(object):
        pass

    scene.SceneCanvas = SceneCanvas
    scene.visuals = mock.MagicMock()
    scene.cameras = mock.MagicMock()
    vispy.scene = scene
    vispy.app = mock.MagicMock()
    visuals = types.ModuleType("vispy.visuals")
    transforms = types.ModuleType("vispy.visuals.transforms")
    transforms.STTransform = mock.MagicMock()
    visuals.transforms = transforms
    vispy.visuals = visuals
    sys.modules.update({"vispy": vispy, "vispy.scene": scene, "vispy.visuals": visuals,
                        "vispy.visuals.transforms": transforms})
    np.float = float  # removed numpy aliases the reference still uses
    np.int = int
    collections.Iterable = collections.abc.Iterable
    import omnistereo.gum as gum
    import omnistereo.panorama as pano
    import omnistereo.camera_models as cm
    import omnistereo.common_cv as ccv
    import omnistereo.pose_est_tools as pet
    return gum, pano, cm, ccv, pet


def build_gums(gum, pano, scale=1.0, width=1200):
    c = np.array([319.5, 239.5]) * scale + (scale - 1.0) * 0.5
    size = (int(640 * scale), int(480 * scale))
    top = gum.GUM("/nonexistent", z_axis=+1.0, image_size_pixels=size, center_uv_point=c, xi3=+0.9,
                  gamma1=150. * scale, gamma2=150. * scale)
    bot = gum.GUM("/nonexistent", z_axis=-1.0, image_size_pixels=size, center_uv_point=c, xi3=-0.9,
                  gamma1=60. * scale, gamma2=60. * scale)
    top.F[2, 0] = 150.
    bot.F[2, 0] = 50.
    top.set_pose(top.F[:3, 0], np.identity(3))
    bot.set_pose(bot.F[:3, 0], np.identity(3))
    gs = gum.GUMStereo(top, bot, center_point_top=c, center_point_top_inner=c, center_point_top_outer=c,
                       center_point_bottom=c, center_point_bottom_inner=c, center_point_bottom_outer=c,
                       radius_top_outer=226 * scale, radius_top_inner=113 * scale,
                       radius_bottom_outer=101 * scale, radius_bottom_inner=50 * scale)
    top.panorama = pano.Panorama(top, width=width)
    bot.panorama = pano.Panorama(bot, width=width)
    return gs, top, bot


def model_params(m):
    pp = m.precalib_params
    return np.array([float(np.ravel(m.Cp_wrt_M)[0]), float(np.ravel(m.Cp_wrt_M)[1]), float(np.ravel(m.Cp_wrt_M)[2]),
                     pp.k1, pp.k2, pp.k3, pp.gamma1, pp.gamma2, pp.alpha_c, pp.u_center, pp.v_center,
                     float(m.z_axis), float(bool(pp.use_distortion))], dtype=np.float64)


def main():
    gum, pano, cm, ccv, pet = import_reference()
    rng = np.random.default_rng(20261003)
    out = {}
    gs, top, bot = build_gums(gum, pano)
    rows_sel = np.array([0, 1, 2, 10, 59, 60, 61, 119, 120, 121])
    for name, m in (("top", top), ("bot", bot)):
        pn = m.panorama
        out[name + "_params"] = model_params(m)
        out[name + "_F"] = np.asarray(m.F[:3, 0], dtype=np.float64)
        out[name + "_T_model_wrt_C"] = np.asarray(m.T_model_wrt_C, dtype=np.float64)
        out[name + "_elev"] = np.array([m.lowest_elevation_angle, m.highest_elevation_angle,
                                        m.globally_lowest_elevation_angle, m.globally_highest_elevation_angle])
        out[name + "_radii"] = np.array([m.inner_img_radius, m.outer_img_radius], dtype=np.float64)
        out[name + "_pano"] = np.array([pn.rows, pn.cols, pn.pixel_size, pn.cyl_height_max, pn.z_height_min,
                                        pn.cyl_circumference])
        out[name + "_lut_rows"] = rows_sel
        out[name + "_lut_x"] = np.asarray(pn.world2cam_LUT_map_x)[rows_sel]
        out[name + "_lut_y"] = np.asarray(pn.world2cam_LUT_map_y)[rows_sel]
        out[name + "_lut_nan_count"] = np.array([int(np.isnan(pn.world2cam_LUT_map_x).sum())])
        out[name + "_lut_x_f32_sum"] = np.array([np.nansum(np.asarray(pn.world2cam_LUT_map_x).astype(np.float32),
                                                          dtype=np.float64)])
    out["units"] = np.array([gs.units])

    # a7 / a8: pano pixels (some outside the image -> NaN) -> angles -> bearings
    n = 400
    m_top = np.ones((n, 3))
    m_top[:, 0] = rng.uniform(-5, 1205, n)
    m_top[:, 1] = rng.uniform(-3, 125, n)
    m_top[:8, 0] = [0.0, 1199.0, 1199.999, 1200.0, -0.0, 600.5, 100.25, 700.0]
    m_top[:8, 1] = [0.0, 121.0, 121.999, 122.0, 60.0, -1e-9, 30.5, 80.0]
    m_bot = m_top.copy()
    m_bot[:, 0] += rng.uniform(-2.5, 2.5, n)
    m_bot[:, 1] -= rng.uniform(0.5, 20, n)
    out["m_top"], out["m_bot"] = m_top, m_bot
    az1, el1 = top.panorama.get_direction_angles_from_pixel_pano(m_top, use_LUTs=False)
    az2, el2 = bot.panorama.get_direction_angles_from_pixel_pano(m_bot, use_LUTs=False)
    out["az_top"], out["el_top"], out["az_bot"], out["el_bot"] = az1, el1, az2, el2
    out["bearing_top"] = top.get_3D_point_from_angles_wrt_focus(azimuth=az1, elevation=el1)[0]
    out["bearing_bot"] = bot.get_3D_point_from_angles_wrt_focus(azimuth=az2, elevation=el2)[0]
    # a9 / a10
    with np.errstate(all="ignore"):
        X = gs.get_triangulated_point_from_direction_angles(dir_angs_top=(az1, el1), dir_angs_bot=(az2, el2),
                                                            use_midpoint_triangulation=True)[0]
        out["tri_X_homo"] = X
        out["range_ok_500_7000"] = gs.filter_panoramic_points_due_to_range(X, min_3D_range=500.0, max_3D_range=7000.0)
        out["range_ok_min_only"] = gs.filter_panoramic_points_due_to_range(X, min_3D_range=900.0, max_3D_range=0.0)
    # a6 gates
    out["gate_stereo"] = ccv.filter_pixel_correspondences(matched_points_top=m_top[:, :2], matched_points_bot=m_bot[:, :2],
                                                          min_rectified_disparity=1, max_horizontal_diff=2.5)
    out["gate_f2f"] = ccv.filter_pixel_correspondences(matched_points_top=m_top[:, :2], matched_points_bot=m_bot[:, :2],
                                                       min_rectified_disparity=-1, max_horizontal_diff=1.25)
    # world point round trip
    Pw = np.array([[[2000.0, 500.0, 300.0, 1.0]]])
    out["roundtrip_point"] = Pw[0, 0, :3]
    u_t, v_t, _ = top.get_pixel_from_3D_point_wrt_C(Pw)
    u_b, v_b, _ = bot.get_pixel_from_3D_point_wrt_C(Pw)
    out["roundtrip_px"] = np.array([u_t.item(), v_t.item(), u_b.item(), v_b.item()])

    # a12: RGB-D back-projection, both depth encodings
    depth = rng.uniform(0.3, 9.0, (48, 64)).astype(np.float32)
    depth[rng.random((48, 64)) < 0.1] = 0.0
    u = rng.integers(0, 64, 200).astype(np.uint)
    v = rng.integers(0, 48, 200).astype(np.uint)
    out["rgbd_depth"], out["rgbd_u"], out["rgbd_v"] = depth, u.astype(np.int64), v.astype(np.int64)
    for tag, is_z in (("z", True), ("radial", False)):
        cam = cm.RGBDCamModel(fx=554.256258, fy=554.256258, center_x=31.5, center_y=23.5, depth_is_Z=is_z)
        with np.errstate(all="ignore"):
            xyz = cam.get_XYZ(depth=depth, u_coords=u, v_coords=v)
            out["rgbd_xyz_" + tag] = xyz
            good = ~np.isnan(xyz[..., 2])
            out["rgbd_bearing_" + tag] = cm.get_normalized_points(xyz[good])
    out["rgbd_intrinsics"] = np.array([554.256258, 554.256258, 31.5, 23.5, 1.0 / 1000.0])

    # a16
    tr = object.__new__(pet.TrackerSE3)
    out["ransac_iters_3_065"] = np.array([pet.TrackerSE3.compute_num_of_iterations_RANSAC(tr, 3, 0.65)])
    out["ransac_iters_3_090"] = np.array([pet.TrackerSE3.compute_num_of_iterations_RANSAC(tr, 3, 0.90)])
    out["thr_5deg"] = np.array([1.0 - np.cos(np.deg2rad(5.0))])

    # a17: the reference's own restatement of the absolute-pose score (central form)
    ang = 0.07
    axis = np.array([0.2, -0.5, 0.84])
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    t = np.array([35.0, -80.0, 12.5])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    P = rng.normal(size=(64, 3)) * 2500.0
    f = (P - t) @ R
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    f[::3] = rng.normal(size=f[::3].shape)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    scores = pet.get_selected_distances_to_model(T, np.arange(64), P, f, False)
    out["score_T"], out["score_p"], out["score_f"], out["score_expected"] = T[:3], P, f, np.array(scores)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
