"""GPU, BASELINE config 5: the RGB-D (perspective) path -- 640x480 BGR + depth, whole-image GFT (2000 budget) +
ORB descriptors, depth back-projection, frame-to-frame matching, central P3P RANSAC, LM -- batched through the
C ABI (RGBDPairPipeline) against the reference's control flow on the CPU oracle (tests/refflow.py): gray images,
keypoints, descriptors, valid-depth compaction, correspondences and inlier masks bit-exact; FP64 points and
bearings bit-exact (+ - * / sqrt only); refined pose rel-tol 1e-6."""
import numpy as np
import pytest

import refflow
import synth
from vo_single_camera_sos_amd import orb_pattern, synthetic
from vo_single_camera_sos_amd.pipeline import RGBDCamConfig, RGBDPairPipeline

pytestmark = pytest.mark.gpu

FX = FY = 554.256258
CX, CY = 319.5, 239.5


def _render_rgbd(room, R, t, rng, depth_is_Z=True):
    return synthetic.render_rgbd(room, R, t, rng, FX, FY, CX, CY, depth_is_Z=depth_is_Z)


YAW = np.deg2rad(40.0)  # the rig looks into a corner of the room: two walls in view (EPnP needs non-coplanar points)
R_YAW = np.array([[np.cos(YAW), -np.sin(YAW), 0.0], [np.sin(YAW), np.cos(YAW), 0.0], [0.0, 0.0, 1.0]])


@pytest.mark.parametrize("depth_is_Z,thr_deg,algo", [(True, 5.0, "EPNP"), (False, 0.5, "EPNP"), (True, 0.5, "KNEIP"),
                                                     (True, 0.5, "GAO"), (True, 0.5, "GP3P"), (True, 5.0, "TWOPT")])
def test_rgbd_pairs_end_to_end_parity(ctx, depth_is_Z, thr_deg, algo):
    """thr_deg 5 is the reference's RANSAC threshold (pose_est_tools.py:675); with it the narrow-FOV pose is only
    loosely constrained (rotation trades against sideways translation), so the recovery of the planted motion
    is asserted on the 0.5-degree run."""
    B, nfeat = 3, 2000
    with pytest.raises(ValueError):   # an unknown solver name is refused, never mapped to some default
        RGBDPairPipeline(ctx, RGBDCamConfig(fx=FX, fy=FY, center_x=CX, center_y=CY), 1, pose_est_algorithm="P4P")
    bgr, depth, poses = [], [], []
    for i in range(B):
        rng = np.random.default_rng(900 + i)
        room = synthetic.Room(seed=900 + i, half_x=(1800.0, 2600.0), half_y=(2500.0, 3500.0), cells=(150.0, 40.0))
        R, t = synthetic.random_step(rng, max_t=80.0, max_deg=8.0)
        for (Rw, tw) in ((R_YAW, np.zeros(3)), (R_YAW @ R, R_YAW @ t)):   # pose of the current rig in the reference rig: (R, t)
            im, dp = _render_rgbd(room, Rw, tw, rng, depth_is_Z)
            bgr.append(im)
            depth.append(dp)
        poses.append((R, t))
    bgr, depth = np.stack(bgr), np.stack(depth)
    depth[5] = 0.0                                            # pair 2: no depth in the current frame -> cannot track
    cam = RGBDCamConfig(fx=FX, fy=FY, center_x=CX, center_y=CY, depth_is_Z=depth_is_Z, min_range=0.8, max_range=7.0)
    pipe = RGBDPairPipeline(ctx, cam, B, num_of_features=nfeat, max_iter=400, seed=31,
                            thr=1.0 - np.cos(np.deg2rad(thr_deg)), pose_est_algorithm=algo)
    assert pipe.kp_cap > 1024                                 # large-mask detector variant
    pipe.load_frames(bgr, depth)
    pipe.step()
    rec = pipe.results()
    ctx.synchronize()
    assert not pipe.status.cpu().numpy().any()

    # the same step behind ONE C-ABI call (sosvo_rgbd_pair_batch) gives the same records, bit for bit
    from vo_single_camera_sos_amd.pipeline import RGBDPairBatch
    one = RGBDPairBatch(ctx, cam, B, num_of_features=nfeat, max_iter=400, seed=31, thr=1.0 - np.cos(np.deg2rad(thr_deg)),
                        pose_est_algorithm=algo)
    one.load_frames(bgr, depth)
    rec1 = one.step()
    ctx.synchronize()
    assert np.array_equal(rec1.cpu().numpy(), rec.cpu().numpy())

    ca, sa = orb_pattern.angle_cos_sin(orb_pattern.GFT_KEYPOINT_ANGLE)
    rc = refflow.RGBDParams(FX, FY, CX, CY, cam.focal_length_m, depth_is_Z, 0.8, 7.0, cam.f2f_max_hdiff, 1.0)
    frames = [refflow.rgbd_frame(rc, bgr[f], depth[f], nfeat, orb_pattern.orb_pattern(), ca, sa) for f in range(2 * B)]
    g_n, M = pipe.n.cpu().numpy(), pipe.frames["M"].cpu().numpy()
    g = {k: v.cpu().numpy() for k, v in pipe.frames.items()}
    for f, fr in enumerate(frames):
        assert g_n[f] == fr["n_kp"] and M[f] == len(fr["X"]), (f, g_n[f], fr["n_kp"], M[f], len(fr["X"]))
        assert np.array_equal(g["m"][f, : M[f]], fr["m"]) and np.array_equal(g["d"][f, : M[f]], fr["d"])
        assert np.array_equal(g["X"][f, : M[f]], fr["X"]) and np.array_equal(g["b"][f, : M[f]], fr["b"])
    assert g_n[:5].min() > 1500 and M[:5].min() > 1300 and M[5] == 0
    rec = rec.cpu().numpy()
    mask = pipe.ransac["mask"].cpu().numpy()
    cq, ct = pipe.corr["q"].cpu().numpy(), pipe.corr["t"].cpu().numpy()
    for i in range(B):
        w = refflow.track_pair_rgbd(rc, frames[2 * i], frames[2 * i + 1], pipe.thr, 400, seed=31 + i, epnp=(algo == "EPNP"),
                                    gp3p=(algo in ("GAO", "GP3P")), twopt=(algo == "TWOPT"))
        n = len(w["corr"]["q"])
        assert rec[i, 13] == n and np.array_equal(cq[i, :n], w["corr"]["q"]) and np.array_equal(ct[i, :n], w["corr"]["t"])
        assert rec[i, 14] == w["ransac"]["status"] and rec[i, 12] == w["ransac"]["n_inliers"]
        assert rec[i, 15] == w["ransac"]["best_iter"]
        assert np.array_equal(mask[i, :n].astype(bool), w["ransac"]["mask"])
        assert np.allclose(rec[i, :12].reshape(3, 4), w["T"], rtol=1e-6, atol=1e-9)   # north_star's bar for the pose ...
        assert np.array_equal(rec[i, :12].reshape(3, 4), w["T"])                        # ... which is met bit for bit
    assert rec[2, 14] == 1 and rec[2, 13] == 0 and (rec[:2, 14] == 0).all()
    # ("TWOPT" keeps the binding's identity rotation prior: a translation-only model of a rotated camera)
    assert (rec[:2, 12] > (300 if algo != "TWOPT" else 3)).all()
    # the planted motion is recovered: pose of the current camera in the reference camera frame, metres
    C = np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]])
    for i in range(2 if thr_deg < 1.0 and algo != "TWOPT" else 0):
        R, t = poses[i]
        T = rec[i, :12].reshape(3, 4)
        ang, _ = synth.pose_error(T, C.T @ R @ C, C.T @ t * 1e-3)
        assert ang < np.deg2rad(0.5) and np.linalg.norm(T[:, 3] - C.T @ t * 1e-3) < 0.02, (ang, T[:, 3], C.T @ t * 1e-3)
