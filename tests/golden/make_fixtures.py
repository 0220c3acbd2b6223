"""Generates tests/golden/geometry_c2.npz and tests/golden/geometry_distorted.npz by IMPORTING the reference's numpy geometry.

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
OUT_DISTORTED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "geometry_distorted.npz")


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


# ---- a second rig on the branches the first one leaves untouched (missing 5 / next 2a) ----------------
# non-zero radial distortion k1..k3, an off-axis projection point (xi1, xi2), skew alpha_c, gamma1 != gamma2, a principal
# point away from the image centre, and mask centres that differ between the inner and the outer circle of a mirror
# (reference: gum.py:121-140, :169-214, :1368-1385, :2512-2562, :2942-2971; camera_models.py:964-990, :1384-1480)
DISTORTED = dict(
    top=dict(xi1=0.012, xi2=-0.008, xi3=+0.9, gamma1=150.0, gamma2=151.5, alpha_c=0.004, k=(-0.021, 0.0035, -0.0004)),
    bot=dict(xi1=-0.01, xi2=0.006, xi3=-0.9, gamma1=60.0, gamma2=60.7, alpha_c=-0.003, k=(0.015, -0.002, 0.0003)),
    center_uv=(322.25, 236.75),
    center_top=(321.0, 238.0), center_top_inner=(321.0, 238.0), center_top_outer=(322.0, 239.0),
    center_bot=(323.5, 235.5), center_bot_inner=(323.5, 235.5), center_bot_outer=(323.0, 235.0),
    radii=dict(top_outer=226, top_inner=113, bot_outer=101, bot_inner=50), F_top=150.0, F_bot=50.0, width=1200)


def build_gums_distorted(gum, pano):
    D = DISTORTED
    c = np.array(D["center_uv"])
    kw = dict(image_size_pixels=(640, 480), center_uv_point=c)
    models = []
    for name, z in (("top", +1.0), ("bot", -1.0)):
        d = D[name]
        m = gum.GUM("/nonexistent", z_axis=z, xi1=d["xi1"], xi2=d["xi2"], xi3=d["xi3"], gamma1=d["gamma1"], gamma2=d["gamma2"],
                    alpha_c=d["alpha_c"], **kw)
        m.precalib_params.k1, m.precalib_params.k2, m.precalib_params.k3 = d["k"]   # (no keyword for these: gum.py:84-86)
        models.append(m)
    top, bot = models
    top.F[2, 0], bot.F[2, 0] = D["F_top"], D["F_bot"]
    top.set_pose(top.F[:3, 0], np.identity(3))
    bot.set_pose(bot.F[:3, 0], np.identity(3))
    a = lambda k: np.array(D[k])  # noqa: E731
    # the constructor lifts the boundary circles to elevations by optimising over the FORWARD projection
    # (camera_models.py:1312-1382): the limits below are the reference's own numbers for this rig
    gs = gum.GUMStereo(top, bot, center_point_top=a("center_top"), center_point_top_inner=a("center_top_inner"),
                       center_point_top_outer=a("center_top_outer"), center_point_bottom=a("center_bot"),
                       center_point_bottom_inner=a("center_bot_inner"), center_point_bottom_outer=a("center_bot_outer"),
                       radius_top_outer=D["radii"]["top_outer"], radius_top_inner=D["radii"]["top_inner"],
                       radius_bottom_outer=D["radii"]["bot_outer"], radius_bottom_inner=D["radii"]["bot_inner"])
    top.panorama = pano.Panorama(top, width=D["width"])
    bot.panorama = pano.Panorama(bot, width=D["width"])
    return gs, top, bot


def distorted_fixture(gum, pano, cm, ccv):
    rng = np.random.default_rng(20261004)
    gs, top, bot = build_gums_distorted(gum, pano)
    out = {}
    for name, m in (("top", top), ("bot", bot)):
        pn = m.panorama
        pp = m.precalib_params
        out[name + "_params"] = model_params(m)
        out[name + "_F"] = np.asarray(m.F[:3, 0], dtype=np.float64)
        out[name + "_T_model_wrt_C"] = np.asarray(m.T_model_wrt_C, dtype=np.float64)
        out[name + "_elev"] = np.array([m.lowest_elevation_angle, m.highest_elevation_angle,
                                        m.globally_lowest_elevation_angle, m.globally_highest_elevation_angle])
        out[name + "_radii"] = np.array([m.inner_img_radius, m.outer_img_radius], dtype=np.float64)
        out[name + "_centers"] = np.array([pp.center_point, pp.center_point_inner, pp.center_point_outer], dtype=np.float64)
        out[name + "_pano"] = np.array([pn.rows, pn.cols, pn.pixel_size, pn.cyl_height_max, pn.z_height_min,
                                        pn.cyl_circumference])
        lx, ly = np.asarray(pn.world2cam_LUT_map_x), np.asarray(pn.world2cam_LUT_map_y)
        nan_rows = np.flatnonzero(np.isnan(lx).any(axis=1))
        rows_sel = np.unique(np.concatenate([[0, 1, 2, 11, 40, pn.rows // 2, pn.rows - 3, pn.rows - 2, pn.rows - 1],
                                             nan_rows[:2], nan_rows[-2:],
                                             [max(0, nan_rows.min() - 1)] if nan_rows.size else [],
                                             [min(pn.rows - 1, nan_rows.max() + 1)] if nan_rows.size else []]).astype(np.int64))
        out[name + "_lut_rows"] = rows_sel
        out[name + "_lut_x"], out[name + "_lut_y"] = lx[rows_sel], ly[rows_sel]
        out[name + "_lut_nan_count"] = np.array([int(np.isnan(lx).sum())])
        out[name + "_lut_nan_rows"] = nan_rows.astype(np.int64)
        # every entry of the float32 maps (what cv2.remap is given, panorama.py:291-293) through two order-independent sums
        for tag, arr in (("x", lx), ("y", ly)):
            f32 = arr.astype(np.float32)
            out[name + "_lut_%s_f32_sum" % tag] = np.array([np.nansum(f32, dtype=np.float64)])
            w = (np.arange(f32.size, dtype=np.float64).reshape(f32.shape) % 8191.0) + 1.0
            out[name + "_lut_%s_f32_wsum" % tag] = np.array([np.nansum(f32.astype(np.float64) * w)])
    out["units"] = np.array([gs.units])
    # the forward projection itself on scattered 3-D points wrt [C] (distortion, skew and off-axis branch, point by point)
    n = 300
    P = np.ones((1, n, 4))
    P[0, :, :3] = rng.normal(size=(n, 3)) * np.array([2500.0, 2500.0, 700.0]) + np.array([0.0, 0.0, 100.0])
    out["proj_points_wrt_C"] = P[0, :, :3]
    for name, m in (("top", top), ("bot", bot)):
        u, v, _ = m.get_pixel_from_3D_point_wrt_C(P)
        out["proj_uv_" + name] = np.stack([np.asarray(u).reshape(-1), np.asarray(v).reshape(-1)], axis=1)
    # a7 - a10 and the gates on this panorama geometry
    rows, cols = top.panorama.rows, top.panorama.cols
    n = 300
    m_top = np.ones((n, 3))
    m_top[:, 0] = rng.uniform(-5, cols + 5, n)
    m_top[:, 1] = rng.uniform(-3, rows + 3, n)
    m_top[:6, 0] = [0.0, cols - 1.0, cols - 0.001, float(cols), 600.5, 100.25]
    m_top[:6, 1] = [0.0, rows - 1.0, rows - 0.001, float(rows), -1e-9, 30.5]
    m_bot = m_top.copy()
    m_bot[:, 0] += rng.uniform(-2.5, 2.5, n)
    m_bot[:, 1] -= rng.uniform(0.5, 20, n)
    out["m_top"], out["m_bot"] = m_top, m_bot
    az1, el1 = top.panorama.get_direction_angles_from_pixel_pano(m_top, use_LUTs=False)
    az2, el2 = bot.panorama.get_direction_angles_from_pixel_pano(m_bot, use_LUTs=False)
    out["az_top"], out["el_top"], out["az_bot"], out["el_bot"] = az1, el1, az2, el2
    out["bearing_top"] = top.get_3D_point_from_angles_wrt_focus(azimuth=az1, elevation=el1)[0]
    out["bearing_bot"] = bot.get_3D_point_from_angles_wrt_focus(azimuth=az2, elevation=el2)[0]
    with np.errstate(all="ignore"):
        X = gs.get_triangulated_point_from_direction_angles(dir_angs_top=(az1, el1), dir_angs_bot=(az2, el2),
                                                            use_midpoint_triangulation=True)[0]
        out["tri_X_homo"] = X
        out["range_ok_500_7000"] = gs.filter_panoramic_points_due_to_range(X, min_3D_range=500.0, max_3D_range=7000.0)
    out["gate_stereo"] = ccv.filter_pixel_correspondences(matched_points_top=m_top[:, :2], matched_points_bot=m_bot[:, :2],
                                                          min_rectified_disparity=1, max_horizontal_diff=2.5)
    np.savez_compressed(OUT_DISTORTED, **out)
    print("wrote", OUT_DISTORTED, os.path.getsize(OUT_DISTORTED), "bytes;", len(out), "arrays")


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

    if "--distorted-only" not in sys.argv:
        np.savez_compressed(OUT, **out)
        print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")
    distorted_fixture(gum, pano, cm, ccv)


if __name__ == "__main__":
    main()
