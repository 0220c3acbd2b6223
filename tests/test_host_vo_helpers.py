"""CPU: the host helpers of the VO loop against tests/golden/transforms.npz, captured
from the reference's omnistereo.transformations / common_tools by tests/golden/make_transform_fixtures.py; the
sequence readers; the rig JSON round trip.  No GPU."""
import os

import numpy as np
import pytest

from vo_single_camera_sos_amd.omnistereo import common_cv, common_tools as ct, transformations as tr
from vo_single_camera_sos_amd.omnistereo.gum import gums_from_dict, gums_to_dict, synthetic_gums
from vo_single_camera_sos_amd.omnistereo.panorama import Panorama

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transforms.npz"))
TOL = dict(rtol=0, atol=1e-14)


def test_quaternions_and_matrices_match_reference():
    assert np.allclose(np.stack([tr.quaternion_from_matrix(m, isprecise=False) for m in G["T"]]), G["quat"], **TOL)
    assert np.allclose(np.stack([tr.quaternion_from_matrix(m) for m in G["T_noisy"]]), G["quat_noisy"], **TOL)
    assert np.array_equal(np.stack([tr.quaternion_matrix(q) for q in G["quat"]]), G["quat_matrix"])
    assert np.array_equal(np.stack([tr.translation_from_matrix(m) for m in G["T"]]), G["trans"])
    assert np.array_equal(np.stack([tr.inverse_matrix(m) for m in G["T"]]), G["inv"])
    n = len(G["T"])
    cat = np.stack([tr.concatenate_matrices(G["T"][i], G["T"][(i + 1) % n], G["T"][(i + 2) % n]) for i in range(n)])
    assert np.array_equal(cat, G["concat3"])
    assert np.array_equal(np.array([tr.rpe_rotation_metric(m) for m in G["T_noisy"]]), G["rpe_rot"])
    assert np.array_equal(np.array([tr.rpe_translation_metric(m) for m in G["T_noisy"]]), G["rpe_trans"])
    assert (G["quat"][:, 0] >= 0).all() and np.allclose(np.linalg.norm(G["quat"], axis=1), 1.0)


def test_doctest_values_of_the_reference_module():
    # transformations.py doctests: quaternion_from_matrix(identity) = [1, 0, 0, 0]; diag(1, -1, -1, 1) -> [0, 1, 0, 0]
    assert np.allclose(tr.quaternion_from_matrix(np.identity(4), True), [1, 0, 0, 0])
    q = tr.quaternion_from_matrix(np.diag([1.0, -1.0, -1.0, 1.0]))
    assert np.allclose(q, [0, 1, 0, 0]) or np.allclose(q, [0, -1, 0, 0])
    R = tr.rotation_matrix(0.123, [1, 2, 3])
    assert np.allclose(tr.quaternion_from_matrix(R, True), [0.9981095, 0.0164262, 0.0328524, 0.0492786])
    assert np.allclose(tr.quaternion_matrix([0.99810947, 0.06146124, 0, 0]), tr.rotation_matrix(0.123, [1, 0, 0]))
    assert np.allclose(tr.rotation_matrix(np.pi / 2, [0, 0, 1], [1, 0, 0]).dot([0, 0, 0, 1]), [1, -1, 0, 1])


def test_tum_entries_and_pose_file_reader(tmp_path):
    rows = G["tum_rows"]
    assert np.array_equal(np.stack([tr.transform44_from_TUM_entry(list(r), has_timestamp=True) for r in rows]), G["from_tum"])
    got = np.stack([tr.transform44_from_TUM_entry(list(r[1:]), 1000.0, has_timestamp=False) for r in rows])
    assert np.array_equal(got, G["from_tum_scaled"])
    fn = str(tmp_path / "gt_TUM.txt")
    with open(fn, "w") as fh:
        fh.write("# timestamp tx ty tz qx qy qz qw\n")
        for r in rows:
            fh.write(" ".join(repr(float(v)) for v in r) + "\n")
    for zero_up in (0, 1):
        p7, mats = ct.get_poses_from_file(fn, "m", "mm", [], "tum", bool(zero_up))
        assert np.allclose(np.array(p7), G["file_poses7_%d" % zero_up], rtol=0, atol=1e-12)
        assert np.allclose(np.stack(mats), G["file_mats_%d" % zero_up], rtol=0, atol=1e-12)
    assert np.allclose(ct.get_poses_from_file(fn, zero_up_wrt_origin=True)[1][0], np.identity(4), atol=1e-15)
    with pytest.raises(ValueError):
        ct.get_poses_from_file(fn, pose_format="kitti")
    # the POV-Ray layout (common_tools.py:636-705): translation + static x-y-z Euler angles in degrees, comma-separated
    assert np.allclose(np.stack([tr.quaternion_from_euler(*np.deg2rad(r[3:]), "sxyz") for r in G["pov_rows"]]),
                       G["euler_quat_sxyz"], rtol=0, atol=1e-15)
    fp = str(tmp_path / "gt_povray.txt")
    with open(fp, "w") as fh:
        fh.write("# tx, ty, tz, rx, ry, rz\n")
        for r in G["pov_rows"]:
            fh.write(", ".join(repr(float(v)) for v in r) + "\n")
    for zero_up in (0, 1):
        p7, mats = ct.get_poses_from_file(fp, "cm", "mm", [], "povray", bool(zero_up))
        assert np.allclose(np.array(p7), G["pov_poses7_%d" % zero_up], rtol=0, atol=1e-12)
        assert np.allclose(np.stack(mats), G["pov_mats_%d" % zero_up], rtol=0, atol=1e-12)
    units = [ct.get_length_units_conversion_factor(a, b) for a, b in (("mm", "m"), ("m", "mm"), ("cm", "m"), ("m", "m"))]
    assert np.array_equal(units, G["units_mm_m"])
    out = str(tmp_path / "out.txt")
    ct.save_as_tum_poses_to_file(out, [[1.0, 2.0, 3.0, 0.0, 0.0, 0.0, 1.0]], "mm", "tum", "m")
    assert open(out).read() == "0 0.001000000 0.002000000 0.003000000 0.000000000 0.000000000 0.000000000 1.000000000\n"


def test_sequence_readers(tmp_path):
    rng = np.random.default_rng(3)
    d = tmp_path / "omni"
    d.mkdir()
    imgs = {}
    for i in (10, 2, 1):  # written out of order: the listing must come back sorted
        imgs[i] = rng.integers(0, 256, (12, 17, 3), dtype=np.uint8)
        common_cv.imwrite(str(d / ("image-%04d.png" % i)), imgs[i])
    (d / "notes.txt").write_text("x")
    names = common_cv.get_images(str(d / "image-*.png"), return_names_only=True)
    assert [os.path.basename(n) for n in names] == ["image-0001.png", "image-0002.png", "image-0010.png"]
    assert common_cv.get_images(str(d / "image-*.png"), indices_list=[2, 0], return_names_only=True) == [names[2], names[0]]
    got = common_cv.get_images(str(d / "image-*.png"))
    assert np.array_equal(got[0], imgs[1]) and np.array_equal(got[2], imgs[10])   # BGR round trip
    assert common_cv.imread(str(d / "missing.png")) is None
    from PIL import Image
    depth = rng.integers(0, 9000, (12, 17)).astype(np.uint16)
    Image.fromarray(depth).save(str(tmp_path / "d.png"))
    f = common_cv.get_depthmap_float32_from_png(str(tmp_path / "d.png"), 1.0 / 1000.0)
    assert f.dtype == np.float32 and np.array_equal(f, np.float32((1.0 / 1000.0) * depth))


def test_rig_json_round_trip():
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=1200)
    back = gums_from_dict(gums_to_dict(gs))
    for a, b in ((gs.top_model, back.top_model), (gs.bot_model, back.bot_model)):
        assert a.precalib_params.__dict__ == b.precalib_params.__dict__
        assert np.array_equal(a.F, b.F) and a.image_size == b.image_size and a.z_axis == b.z_axis
        for x, y in zip(a.panorama.float32_maps(), b.panorama.float32_maps()):
            assert np.array_equal(x, y, equal_nan=True)
    assert back.units == gs.units and back.top_model.panorama.cols == 1200
    with pytest.raises(ValueError):
        gums_from_dict({"format": "pickle"})


def _load_exporter():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("export_reference_gums", os.path.join(root, "scripts", "export_reference_gums.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_side_export_of_a_calibrated_rig(tmp_path):
    """scripts/export_reference_gums.py (INTEGRATION.md "Bringing a calibrated rig over") on an object that carries the
    REFERENCE's attribute names (gum.py:77-116, :169-214; camera_models.py:884-936, :972-988) with the numbers of the
    reference-generated distorted rig: the JSON it writes loads into a rig whose unwrap LUT equals the reference's."""
    import json
    from types import SimpleNamespace as NS
    from distorted_rig import D
    from vo_single_camera_sos_amd.omnistereo.gum import load_gums_json
    exp = _load_exporter()

    def ref_like(name):
        xi1, xi2, xi3, k1, k2, k3, g1, g2, alpha, uc, vc, z_axis, use_d = D[name + "_params"]
        c, c_in, c_out = D[name + "_centers"]
        pp = NS(xi1=xi1, xi2=xi2, xi3=xi3, k1=k1, k2=k2, k3=k3, p1=0.0, p2=0.0, gamma1=g1, gamma2=g2, alpha_c=alpha, u_center=uc,
                v_center=vc, use_distortion=bool(use_d), center_point=np.array(c), center_point_inner=np.array(c_in),
                center_point_outer=np.array(c_out), image_size=np.array([640, 480]))
        F = np.ones((4, 1))
        F[:3, 0] = D[name + "_F"]
        return NS(precalib_params=pp, Cp_wrt_M=np.array([xi1, xi2, xi3]), z_axis=z_axis, F=F,
                  lowest_elevation_angle=D[name + "_elev"][0], highest_elevation_angle=D[name + "_elev"][1],
                  inner_img_radius=D[name + "_radii"][0], outer_img_radius=D[name + "_radii"][1],
                  T_model_wrt_C=D[name + "_T_model_wrt_C"], panorama=NS(cols=int(D[name + "_pano"][1])))
    gums = NS(top_model=ref_like("top"), bot_model=ref_like("bot"), units=str(D["units"][0]))
    doc = exp.gums_to_sosvo_json(gums)
    assert doc["format"] == "sosvo-gums-1" and doc["panorama_width"] == 1200 and "R_model_wrt_C" not in doc["top"]
    f = str(tmp_path / "rig.json")
    with open(f, "w") as fh:
        json.dump(doc, fh)
    rig = load_gums_json(f)
    for name, m in (("top", rig.top_model), ("bot", rig.bot_model)):
        sel = D[name + "_lut_rows"]
        assert np.array_equal(np.asarray(m.panorama.world2cam_LUT_map_x)[sel], D[name + "_lut_x"], equal_nan=True)
        assert np.array_equal(np.asarray(m.panorama.world2cam_LUT_map_y)[sel], D[name + "_lut_y"], equal_nan=True)
        assert np.array_equal(m.center_point_outer, D[name + "_centers"][2]) and np.array_equal(m.F[:3, 0], D[name + "_F"])
        assert [m.globally_lowest_elevation_angle, m.globally_highest_elevation_angle] == list(D[name + "_elev"][2:])
    # a rotated mirror frame travels as R_model_wrt_C
    a = np.deg2rad(1.5)
    R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
    gums.bot_model.T_model_wrt_C = np.array(gums.bot_model.T_model_wrt_C)
    gums.bot_model.T_model_wrt_C[:3, :3] = R
    doc2 = exp.gums_to_sosvo_json(gums, panorama_width=900)
    assert doc2["panorama_width"] == 900
    from vo_single_camera_sos_amd.omnistereo.gum import gums_from_dict, gums_to_dict
    rig2 = gums_from_dict(doc2, with_panoramas=False)
    assert np.array_equal(rig2.bot_model.T_model_wrt_C[:3, :3], R) and np.array_equal(rig2.top_model.T_model_wrt_C[:3, :3], np.identity(3))
    assert gums_to_dict(rig2)["bottom"]["R_model_wrt_C"] == doc2["bottom"]["R_model_wrt_C"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/omnistereo"), reason="needs the reference tree (build container only)")
def test_reference_side_export_on_a_live_reference_object(tmp_path):
    """The same exporter on a LIVE GUMStereo of the imported reference (the object a user's pickle holds): every LUT entry of
    the loaded rig equals the reference's own Panorama LUT.  Runs in a child interpreter (the reference needs placeholder
    cv2 / pyopengv / vispy modules, tests/golden/make_fixtures.py) so that nothing leaks into this process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import json, sys, os
import numpy as np
sys.path.insert(0, os.path.join(%r, "tests", "golden")); sys.path.insert(0, os.path.join(%r, "scripts"))
import make_fixtures as mf
import export_reference_gums as exp
gum, pano, cm, ccv, pet = mf.import_reference()
gs, top, bot = mf.build_gums_distorted(gum, pano)
json.dump(exp.gums_to_sosvo_json(gs), open(sys.argv[1], "w"))
np.savez(sys.argv[2], tx=top.panorama.world2cam_LUT_map_x, ty=top.panorama.world2cam_LUT_map_y,
         bx=bot.panorama.world2cam_LUT_map_x, by=bot.panorama.world2cam_LUT_map_y)
""" % (root, root)
    f, z = str(tmp_path / "rig.json"), str(tmp_path / "lut.npz")
    env = dict(os.environ, PYTHONPATH="/root/reference", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code, f, z], capture_output=True, text=True, env=env, timeout=900,
                       cwd=str(tmp_path))   # (not the repository root: its `omnistereo` alias package would shadow the reference's)
    assert r.returncode == 0, r.stderr[-2000:]
    from vo_single_camera_sos_amd.omnistereo.gum import load_gums_json
    rig, L = load_gums_json(f), np.load(z)
    for m, kx, ky in ((rig.top_model, "tx", "ty"), (rig.bot_model, "bx", "by")):
        assert np.array_equal(np.asarray(m.panorama.world2cam_LUT_map_x), L[kx], equal_nan=True)
        assert np.array_equal(np.asarray(m.panorama.world2cam_LUT_map_y), L[ky], equal_nan=True)


def test_precalibration_bin_layout(tmp_path):
    """gum.py:216-272: the toolbox's .bin is a run of float64 -- xi (3 or 1), k1 k2 p1 p2 k3, gamma1 gamma2 u v alpha,
    ROI -- with the centre and ROI 1-based; checked against bytes packed by hand."""
    import struct
    from vo_single_camera_sos_amd.omnistereo.gum import GUMParams
    vals = [0.01, -0.02, 0.93, 0.1, -0.05, 0.001, 0.002, 0.003, 151.5, 150.25, 320.5, 240.5, 0.0005, 11.0, 21.0, 631.0, 471.0]
    fn = str(tmp_path / "new.bin")
    with open(fn, "wb") as f:
        f.write(struct.pack("17d", *vals))
    p = GUMParams.from_precalibration_bin(fn, new_method=True)
    assert (p.xi1, p.xi2, p.xi3) == (0.01, -0.02, 0.93)
    assert (p.k1, p.k2, p.p1, p.p2, p.k3) == (0.1, -0.05, 0.001, 0.002, 0.003)
    assert (p.gamma1, p.gamma2, p.u_center, p.v_center, p.alpha_c) == (151.5, 150.25, 319.5, 239.5, 0.0005)
    assert (p.roi_min_x, p.roi_min_y, p.roi_max_x, p.roi_max_y) == (10.0, 20.0, 630.0, 470.0)
    fn2 = str(tmp_path / "roundtrip.bin")
    p.to_precalibration_bin(fn2)
    assert open(fn2, "rb").read() == open(fn, "rb").read()
    # old method: a single xi3 in front, xi1 = xi2 = 0
    fn3 = str(tmp_path / "old.bin")
    with open(fn3, "wb") as f:
        f.write(struct.pack("15d", *vals[2:]))
    q = GUMParams.from_precalibration_bin(fn3, new_method=False)
    assert (q.xi1, q.xi2, q.xi3, q.k1, q.gamma2, q.v_center) == (0.0, 0.0, 0.93, 0.1, 150.25, 239.5)
    with open(fn3, "ab") as f:
        pass
    import pytest
    with pytest.raises(ValueError):
        GUMParams.from_precalibration_bin(fn3, new_method=True)  # two values short


def test_frame_source_thread_lockstep_and_free_running():
    """webcam_live.FrameSourceThread (the CamAsWorkingThread contract, webcam_live.py:256-291): lock-step hands out
    every frame exactly once and None at the end; free-running keeps only the most recent frame."""
    import time
    from vo_single_camera_sos_amd.omnistereo.webcam_live import FrameSourceThread

    class Cam(object):
        def __init__(self, n):
            self.k, self.n = 0, n

        def get_single_frame(self):
            self.k += 1
            return (self.k <= self.n, self.k)
    t = FrameSourceThread(Cam(9), lockstep=True)
    t.start()
    got = []
    while True:
        f = t.current_frame
        if f is None:
            break
        got.append(f)
    t.quit_flag = True
    t.join()
    assert got == list(range(1, 10)) and t.frames_delivered == 9 and t.cam.show_img is False
    t = FrameSourceThread(Cam(10 ** 9), min_period_s=0.002)
    t.start()
    time.sleep(0.05)
    a = t.current_frame
    time.sleep(0.05)
    b = t.current_frame
    t.quit_flag = True
    t.join()
    assert a is not None and b > a and t.current_frame is None


def test_batched_quaternions_and_pose_line_formatting_equal_the_per_frame_forms():
    """run_VO's sequence mode formats a window's pose lines at once: tr.quaternions_from_matrices must equal
    quaternion_from_matrix row by row BIT FOR BIT, and repr() of the Python floats must print what print() prints for the
    numpy scalars (the reference's `print(idx, t[0], ..., sep=" ")`, pose_est_tools.py:1611)."""
    import io
    from vo_single_camera_sos_amd.omnistereo import transformations as tr
    rng = np.random.default_rng(3)
    Ms = []
    for k in range(300):
        q = rng.normal(size=4)
        T = tr.quaternion_matrix(q / np.linalg.norm(q))
        if k % 3 == 0:
            T[:3, :3] += rng.normal(scale=1e-9, size=(3, 3))     # slightly non-orthonormal, as chained poses are
        T[:3, 3] = rng.normal(size=3) * 10.0 ** rng.integers(-7, 6)
        Ms.append(T)
    Ms += [np.identity(4), np.diag([1.0, -1.0, -1.0, 1.0]), np.diag([-1.0, -1.0, 1.0, 1.0])]
    Ms = np.stack(Ms)
    one = np.stack([tr.quaternion_from_matrix(M, isprecise=False) for M in Ms])
    assert np.array_equal(tr.quaternions_from_matrices(Ms).view(np.uint64), one.view(np.uint64))
    assert tr.quaternions_from_matrices(np.zeros((0, 4, 4))).shape == (0, 4)
    vals = np.concatenate([Ms[:, :3, 3].ravel(), one.ravel(), [1e16, 1e-5, -0.0, 123456789012345680.0, 5e-324, np.nan, np.inf]])
    buf = io.StringIO()
    print(*vals, sep=" ", end="", file=buf)
    assert buf.getvalue() == " ".join(repr(v) for v in vals.tolist())


def test_window_staging_copies_every_frame_with_and_without_the_pool():
    """pipeline._stage_rows: frames of a window into the (pinned) staging array -- one by one for short windows, four numpy
    copies at a time otherwise; casting copies (a float64 depth map into a float32 buffer) included."""
    from vo_single_camera_sos_amd.pipeline import _stage_rows
    rng = np.random.default_rng(5)
    for n in (1, 3, 4, 9):
        frames = [rng.integers(0, 255, (6, 8, 3), dtype=np.uint8) for _ in range(n)]
        dst = np.zeros((12, 6, 8, 3), np.uint8)
        _stage_rows(dst, frames)
        assert all(np.array_equal(dst[i], frames[i]) for i in range(n)) and not dst[n:].any()
    depth = [rng.random((6, 8)) for _ in range(5)]
    dst = np.zeros((5, 6, 8), np.float32)
    _stage_rows(dst, depth)
    assert all(np.array_equal(dst[i], depth[i].astype(np.float32)) for i in range(5))
