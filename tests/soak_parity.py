"""Soak run of the WHOLE hot path against the CPU oracle at the bench configuration (1440 x 146 panoramas, ~2000
keypoints per view, 2000 RANSAC iterations): N rendered frame pairs through OverlappedFramePairs on the GPU and
through the reference's control flow on the C oracle (tests/refflow.py, multi-process), every record compared --
counts, status, best iteration exactly, refined pose at rel-tol 1e-6.  A tool, not part of the test suite:

    python tests/soak_parity.py --pairs 64 [--seed 7] [--workers 16]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (this file lives in tests/: it drives the CPU oracle, which only tests may)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--pano-width", type=int, default=1440)
    ap.add_argument("--detector", default="GFT", choices=["GFT", "ORB", "FAST", "AGAST"])
    ap.add_argument("--kp-cap", type=int, default=512, help="keypoint capacity per (frame, mirror, mask)")
    ap.add_argument("--frame-cap", type=int, default=2048,
                    help="stereo points per frame / correspondences per pair the engine holds (the oracle has no such limit: a "
                         "configuration that exceeds it differs by construction -- wide panoramas with OpenCV's BRIEF table do)")
    ap.add_argument("--median", type=int, default=11, help="median window (11 = the SOS frames' setting; 0 = none)")
    ap.add_argument("--features", type=int, default=1000, help="detector budget per azimuthal mask")
    ap.add_argument("--solver", default="P3P", choices=["P3P", "GP3P"],
                    help="hypothesis generator of the non-central RANSAC: one-mirror P3P (bench default) or the generalised P3P")
    ap.add_argument("--rgbd", choices=["EPNP", "KNEIP"], default=None,
                    help="soak the RGB-D path (BASELINE config 5: sosvo_rgbd_pair_batch) with this pose algorithm instead")
    args = ap.parse_args()
    if args.rgbd:
        return main_rgbd(args)
    import multiprocessing
    import refflow
    from vo_single_camera_sos_amd import orb_pattern, synthetic
    from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
    from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=args.pano_width)
    gs.make_annulus_masks((480, 640))
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    rig_kw = dict(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0], min_range=500.0,
                  max_range=7000.0, stereo_min_disp=1.0, stereo_max_hdiff=2.5, f2f_max_hdiff=0.125 * 0.5 * pano.cols,
                  pct_good_matches=1.0)
    B = args.pairs
    omni, _ = synthetic.make_frame_pairs(gs, B, seed=args.seed, workers=args.workers)  # before the GPU is touched (fork)
    import torch
    from vo_single_camera_sos_amd.pipeline import OverlappedFramePairs, RigConfig
    eng = OverlappedFramePairs(0, gs, (480, 640), RigConfig(**rig_kw), B, n_streams=2, num_of_features=args.features, kp_cap=args.kp_cap,
                               frame_cap=args.frame_cap, max_iter=args.iters, adaptive=False, seed=args.seed, detection_method=args.detector,
                               ransac_solver=args.solver, median_win_size=args.median)
    assert not any(int(p.fe.status.max().item()) for p in eng.parts) or args.detector != "GFT"
    eng.load_frames(omni)
    eng.step()
    rec = eng.results().cpu().numpy()
    torch.cuda.synchronize()
    model = eng.model
    ca, sa = orb_pattern.angle_cos_sin(-1.0)
    im_kw = dict(map_x=model.map_x.cpu().numpy(), map_y=model.map_y.cpu().numpy(), omni_masks=model.omni_masks.cpu().numpy(),
                 mask_bits=model.mask_bits_host, nmask=model.nmask, max_corners=args.features, pattern=model.pattern_host, cos_a=ca,
                 sin_a=sa, kp_cap=args.kp_cap, method=args.detector, median_ksize=args.median)
    per = -(-B // args.workers)
    jobs = [(rig_kw, im_kw, omni[2 * lo: 2 * min(B, lo + per)], eng.thr, args.iters, args.seed + lo, args.solver == "GP3P")
            for lo in range(0, B, per)]
    t0 = time.perf_counter()
    with multiprocessing.get_context("spawn").Pool(len(jobs)) as pool:
        outs = pool.map(refflow.pairs_records_worker, jobs)
    want = np.concatenate(outs)
    print("oracle: %d pairs in %.1f s on %d processes" % (B, time.perf_counter() - t0, len(jobs)))
    bad = 0
    exact_pose = int(sum(np.array_equal(rec[i, :12], want[i, :12]) for i in range(B)))
    for i in range(B):
        exact = np.array_equal(rec[i, 12:], want[i, 12:])
        close = np.allclose(rec[i, :12], want[i, :12], rtol=1e-6, atol=1e-9)
        if not (exact and close):
            bad += 1
            print("pair %d differs: gpu %s\n               cpu %s" % (i, rec[i], want[i]))
    n_kp = np.concatenate([p.fe.n.cpu().numpy().reshape(2, -1, model.nmask).sum(-1) for p in eng.parts], axis=1)
    print("detector %s (median %d, %d per mask: %.0f keypoints per view), solver %s:"
          % (args.detector, args.median, args.features, n_kp.mean(), args.solver), end=" ")
    print("soak: %d / %d pairs identical (counts, status, winning iteration exact; pose rel-tol 1e-6); %d / %d refined poses "
          "bit-identical; inliers %.0f mean" % (B - bad, B, exact_pose, B, rec[:, 12].mean()))
    eng.close()
    sys.exit(1 if bad else 0)


def main_rgbd(args):
    import multiprocessing
    import refflow
    from vo_single_camera_sos_amd import synthetic
    B = args.pairs
    fx = fy = 554.256258
    bgr, depth = [], []
    for i in range(B):
        rng = np.random.default_rng(args.seed + i)
        room = synthetic.Room(seed=args.seed + i, half_x=(1800.0, 2600.0), half_y=(2500.0, 3500.0), cells=(150.0, 40.0), yaw_deg=40.0)
        R, t = synthetic.random_step(rng, max_t=80.0, max_deg=8.0)
        for (Rw, tw) in ((np.eye(3), np.zeros(3)), (R, t)):
            im, dp = synthetic.render_rgbd(room, Rw, tw, rng, depth_is_Z=True)
            bgr.append(im)
            depth.append(dp)
    bgr, depth = np.stack(bgr), np.stack(depth)
    import torch
    from vo_single_camera_sos_amd.device import Context
    from vo_single_camera_sos_amd.pipeline import RGBDCamConfig, RGBDPairBatch
    ctx = Context(0)
    cam = RGBDCamConfig(fx=fx, fy=fy, center_x=319.5, center_y=239.5, depth_is_Z=True, min_range=0.8, max_range=7.0)
    one = RGBDPairBatch(ctx, cam, B, num_of_features=2000, max_iter=args.iters, seed=args.seed, pose_est_algorithm=args.rgbd)
    one.load_frames(bgr, depth)
    rec = one.step().cpu().numpy()
    torch.cuda.synchronize()
    cam_kw = dict(fx=fx, fy=fy, cx=319.5, cy=239.5, focal_length_m=cam.focal_length_m, depth_is_Z=True, min_range=0.8,
                  max_range=7.0, f2f_max_hdiff=cam.f2f_max_hdiff, pct_good_matches=1.0)
    per = -(-B // args.workers)
    jobs = [(cam_kw, bgr[2 * lo: 2 * min(B, lo + per)], depth[2 * lo: 2 * min(B, lo + per)], 2000, one.thr, args.iters,
             args.seed + lo, args.rgbd == "EPNP") for lo in range(0, B, per)]
    t0 = time.perf_counter()
    with multiprocessing.get_context("spawn").Pool(len(jobs)) as pool:
        want = np.concatenate(pool.map(refflow.rgbd_pairs_records_worker, jobs))
    print("oracle: %d RGB-D pairs in %.1f s on %d processes" % (B, time.perf_counter() - t0, len(jobs)))
    exact = int(sum(np.array_equal(rec[i], want[i]) for i in range(B)))
    for i in range(B):
        if not np.array_equal(rec[i], want[i]):
            print("pair %d differs: gpu %s\n               cpu %s" % (i, rec[i], want[i]))
    print("soak (RGB-D, %s): %d / %d records bit-identical (pose, counts, status, winning iteration); inliers %.0f mean, %d tracked"
          % (args.rgbd, exact, B, rec[:, 12].mean(), int((rec[:, 14] == 0).sum())))
    ctx.close()
    sys.exit(0 if exact == B else 1)


if __name__ == "__main__":
    main()
