"""GPU: the complete hot path from omni frames to refined pose (ImageFrontEnd + FramePairPipeline) against
the reference's control flow on the CPU oracle (tests/refflow.py) on the same rendered frames:
panoramas, gray images, keypoints, descriptors, stereo correspondences, frame-to-frame correspondences and
inlier masks bit-exact; refined pose rel-tol 1e-6 (BASELINE north_star)."""
import numpy as np
import pytest

import oracle
import refflow
import synth
from vo_single_camera_sos_amd import orb_pattern, synthetic
from vo_single_camera_sos_amd.frontend import DeviceImageModel, ImageFrontEnd
from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
from vo_single_camera_sos_amd.pipeline import FramePairPipeline, RigConfig

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method,nfeat", [("GFT", 330), ("ORB", 250), ("FAST", 250), ("AGAST", 250)])
def test_full_hot_path_from_images(ctx, method, nfeat):
    B = 3
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=1200)
    gs.make_annulus_masks((480, 640))
    omni, poses = synthetic.make_frame_pairs(gs, B, seed=99)
    omni[5] = 0  # a black current frame: nothing to detect -> tracking of pair 2 must fail cleanly
    model = DeviceImageModel(ctx, gs, (480, 640))
    assert model.nmask == 12
    fe = ImageFrontEnd(ctx, model, 2 * B, detection_method=method, num_of_features=nfeat)
    pano = gs.top_model.panorama
    rig_kw = dict(pano_top=(pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max),
                  pano_bot=(pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max),
                  F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0], min_range=500.0, max_range=7000.0,
                  stereo_min_disp=1.0, stereo_max_hdiff=2.5, f2f_max_hdiff=0.125 * 0.5 * pano.cols,
                  pct_good_matches=1.0)
    assert rig_kw["pano_top"] == synth.PANO_C2
    pipe = FramePairPipeline(ctx, RigConfig(**rig_kw), B, frame_cap=2048, max_iter=500, seed=5, front_end=fe)
    fe.load_frames(omni)
    pipe.step()
    rec = pipe.results()
    ctx.synchronize()

    # ---- model constants: the GPU-built azimuthal masks equal masks built with the oracle's unwrap
    mx, my = model.map_x.cpu().numpy(), model.map_y.cpu().numpy()
    for vi, m in enumerate((gs.top_model, gs.bot_model)):
        gpu_bits = m.panorama.mask_bits()
        m.panorama.generate_azimuthal_masks(
            30, 0, elev_mask_padding=10, stand_masks_azimuth_coord_in_degrees_list=[50, 170, 290],
            stand_masks_width_in_degrees=10,
            unwrap_fn=lambda om, vi=vi: oracle.unwrap(np.repeat(om[..., None], 3, 2), None, mx[vi], my[vi])[..., 0])
        assert np.array_equal(m.panorama.mask_bits(), gpu_bits)
        assert np.array_equal(gpu_bits, model.mask_bits_host[vi])

    ca, sa = orb_pattern.angle_cos_sin(-1.0)
    im = refflow.ImageModel(mx, my, model.omni_masks.cpu().numpy(), model.mask_bits_host, 12, nfeat,
                            orb_pattern.orb_pattern(), ca, sa, method=method, kp_cap=fe.kp_cap)
    rp = refflow.RigParams(**rig_kw)
    g_pano, g_gray = fe.pano.cpu().numpy(), fe.gray.cpu().numpy()
    g_kp, g_n, g_desc = fe.kp.cpu().numpy(), fe.n.cpu().numpy(), fe.desc.cpu().numpy()
    assert not fe.status.cpu().numpy().any()
    F = 2 * B
    frames = []
    for f in range(F):
        per_view = []
        for v in range(2):
            kps, descs, pano_o, gray_o = refflow.detect_view(im, omni[f], v)
            assert np.array_equal(g_pano[v, f], pano_o), ("pano", f, v)
            assert np.array_equal(g_gray[v * F + f], gray_o), ("gray", f, v)
            for m in range(12):
                p = (v * F + f) * 12 + m
                assert g_n[p] == len(kps[m]), ("count", f, v, m)
                assert np.array_equal(g_kp[p, : g_n[p]], kps[m]), ("kp", f, v, m)
                assert np.array_equal(g_desc[p, : g_n[p]], descs[m]), ("desc", f, v, m)
            per_view.append((kps, descs))
        frames.append(refflow.stereo_frame(rp, per_view[0][0], per_view[1][0], per_view[0][1], per_view[1][1]))
    M = pipe.frames["M"].cpu().numpy()
    assert [int(x) for x in M] == [len(fr["X"]) for fr in frames]
    # FAST finds few corners on the 11x11-median-blurred panorama: ORB yields far fewer points than GFT here
    assert M[:5].min() > (300 if method == "GFT" else 3) and M[5] == 0
    rec = rec.cpu().numpy()
    mask = pipe.ransac["mask"].cpu().numpy()
    for i in range(B):
        w = refflow.track_pair(rp, frames[2 * i], frames[2 * i + 1], pipe.thr, 500, seed=5 + i)
        n = len(w["corr"]["cam"])
        assert rec[i, 13] == n and rec[i, 14] == w["ransac"]["status"] and rec[i, 12] == w["ransac"]["n_inliers"]
        assert np.array_equal(mask[i, :n].astype(bool), w["ransac"]["mask"])
        assert np.allclose(rec[i, :12].reshape(3, 4), w["T"], rtol=1e-6, atol=1e-9)   # north_star's bar for the pose ...
        assert np.array_equal(rec[i, :12].reshape(3, 4), w["T"])                        # ... which is met bit for bit
    assert rec[2, 14] == 1 and rec[2, 13] == 0           # black frame: no correspondences, status "no model"
    if method == "GFT":
        # the reference's own choice for the non-central RANSAC, generalised P3P on samples across both mirrors
        # (pose_est_tools.py:696), on the same keypoints: every record equals the oracle flow with gp3p hypotheses
        pipe_g = FramePairPipeline(ctx, RigConfig(**rig_kw), B, frame_cap=2048, max_iter=500, seed=5, front_end=fe,
                                   ransac_solver="GP3P")
        pipe_g.stereo()
        pipe_g.track()
        rec_g = pipe_g.results().cpu().numpy()
        mask_g = pipe_g.ransac["mask"].cpu().numpy()
        differs = 0
        for i in range(B):
            w = refflow.track_pair(rp, frames[2 * i], frames[2 * i + 1], pipe.thr, 500, seed=5 + i, gp3p=True)
            n = len(w["corr"]["cam"])
            assert rec_g[i, 13] == n and rec_g[i, 14] == w["ransac"]["status"] and rec_g[i, 12] == w["ransac"]["n_inliers"]
            assert rec_g[i, 15] == w["ransac"]["best_iter"]
            assert np.array_equal(mask_g[i, :n].astype(bool), w["ransac"]["mask"])
            assert np.array_equal(rec_g[i, :12].reshape(3, 4), w["T"])
            differs += int(rec_g[i, 15] != rec[i, 15])
        assert differs > 0                                # a different hypothesis set than the one-mirror mode
        for i in range(2):
            R, t = poses[i]
            assert synth.pose_error(rec_g[i, :12].reshape(3, 4), R, t)[0] < np.deg2rad(2.0)
    for i in range(2 if method == "GFT" else 0):          # the planted motion is recovered (loosely: 5 deg threshold)
        R, t = poses[i]
        ang, _ = synth.pose_error(rec[i, :12].reshape(3, 4), R, t)
        assert ang < np.deg2rad(2.0)


@pytest.mark.parametrize("solver", ["P3P", "GP3P"])
def test_full_hot_path_accuracy_with_a_tight_ransac_threshold(ctx, solver):
    """Outside evidence for the absolute-pose solvers (their arithmetic is unpinned by the reference): the same hot path
    with the RANSAC threshold at 0.5 degrees instead of the reference's 5 (pose_est_tools.py:675-676 -- at 5 degrees the
    LM set keeps correspondences with degrees of back-projection error and the pose is only good to ~0.7 degrees / 4 cm,
    profiles/round3/accuracy_sos.json) must RECOVER the planted motion: every pair within 0.3 degrees and 1 cm, for the
    one-mirror P3P and for the generalised P3P hypotheses.  A wrong-but-self-consistent solver cannot pass this."""
    B = 6
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=1200)
    gs.make_annulus_masks((480, 640))
    omni, poses = synthetic.make_frame_pairs(gs, B, seed=4100)
    model = DeviceImageModel(ctx, gs, (480, 640))
    fe = ImageFrontEnd(ctx, model, 2 * B, detection_method="GFT", num_of_features=1000, kp_cap=512)
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    rig_kw = dict(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0], min_range=500.0,
                  max_range=7000.0, stereo_min_disp=1.0, stereo_max_hdiff=2.5, f2f_max_hdiff=0.125 * 0.5 * pano.cols,
                  pct_good_matches=1.0)
    pipe = FramePairPipeline(ctx, RigConfig(**rig_kw), B, frame_cap=2048, max_iter=2000, seed=11, front_end=fe,
                             thr=1.0 - np.cos(np.deg2rad(0.5)), ransac_solver=solver)
    fe.load_frames(omni)
    pipe.step()
    rec = pipe.results().cpu().numpy()
    assert (rec[:, 14] == 0).all() and (rec[:, 12] >= 150).all(), rec[:, 12:15]
    for i, (R, t) in enumerate(poses):
        T = rec[i, :12].reshape(3, 4)
        ang, _ = synth.pose_error(T, R, t)
        assert np.degrees(ang) < 0.3, (i, np.degrees(ang))
        assert np.linalg.norm(T[:, 3] - t) < 10.0, (i, np.linalg.norm(T[:, 3] - t))   # model units: mm
