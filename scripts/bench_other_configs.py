"""Throughput of the BASELINE configurations that are parity-test cases rather than bench lines (bench.py measures C2):
  C3  1280x960 omni frames -> two 2400x244 panoramas, ~8000 keypoints per view: image front end (K1-K6), per-bucket
      2-NN Hamming matching + sort, stereo gates + bearings + midpoint triangulation (batched), frames/s
  C5  640x480 RGB-D frame pairs, ~2000 keypoints, central RANSAC (EPNP and KNEIP) + LM, one C-ABI call per step, pairs/s
Both split their batch over HIP streams (one libsosvo context each), as the C2 engine does.
One JSON line each; inputs resident in HBM, synthetic.   python scripts/bench_other_configs.py [--frames 192] [--pairs 128]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def timed(fn, sync, warmup=2, steps=10):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=192)
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--only", default="", help="C3 or C5: run just that configuration")
    ap.add_argument("--c3-streams", type=int, default=3, help="HIP streams the C3 batch is split over")
    ap.add_argument("--c5-streams", type=int, default=3, help="one-call RGB-D batches of --pairs pairs run side by side on this many HIP streams")
    args = ap.parse_args()
    from vo_single_camera_sos_amd import synthetic
    from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
    from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
    # ---- render on the host first (C3: a few distinct frames tiled over the batch; C5: distinct pairs)
    gs = synthetic_gums(scale=2.0)
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=2400)
    gs.make_annulus_masks((960, 1280))
    uniq = 4
    omni3 = None if args.only == "C5" else np.stack([synthetic.render_omni(gs, synthetic.Room(seed=70 + i, cells=(180.0, 45.0)), np.eye(3), np.zeros(3), 2.0,
                                            np.random.default_rng(70 + i)) for i in range(uniq)])
    if omni3 is not None:
        omni3 = np.concatenate([omni3] * (-(-args.frames // uniq)))[: args.frames]
    B = args.pairs
    bgr, depth = [], []
    for i in range(min(B, 16)):
        rng = np.random.default_rng(900 + i)
        room = synthetic.Room(seed=900 + i, half_x=(1800.0, 2600.0), half_y=(2500.0, 3500.0), cells=(150.0, 40.0), yaw_deg=40.0)
        R, t = synthetic.random_step(rng, max_t=80.0, max_deg=8.0)
        for (Rw, tw) in ((np.eye(3), np.zeros(3)), (R, t)):
            im, dp = synthetic.render_rgbd(room, Rw, tw, rng, depth_is_Z=True)
            bgr.append(im)
            depth.append(dp)
    reps = -(-B // min(B, 16))
    bgr, depth = np.concatenate([np.stack(bgr)] * reps)[: 2 * B], np.concatenate([np.stack(depth)] * reps)[: 2 * B]

    import torch
    from vo_single_camera_sos_amd.device import Context
    from vo_single_camera_sos_amd.frontend import DeviceImageModel, ImageFrontEnd
    from vo_single_camera_sos_amd.pipeline import FramePairPipeline, RGBDCamConfig, RGBDPairBatch, RigConfig
    ctx = Context(0)
    if args.only != "C5":
        # ---- C3: the batch split over HIP streams (one libsosvo context each), as the C2 engine does: the latency-bound
        # stages of one part (corner selection, the per-bucket 2-NN matching) run under the VALU-bound median of another
        F = args.frames
        S = max(1, min(args.c3_streams, F // 2))
        pano = gs.top_model.panorama
        geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
        rig = RigConfig(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0], min_range=500.0,
                        max_range=7000.0, stereo_min_disp=1.0, stereo_max_hdiff=2.5, f2f_max_hdiff=0.125 * 0.5 * pano.cols,
                        pct_good_matches=1.0)
        model = DeviceImageModel(ctx, gs, (960, 1280))
        model.unwrap_table = ctx.unwrap_prepare(model.omni_masks, model.map_x, model.map_y, (model.H, model.W))
        ctx.synchronize()
        parts = []
        per = -(-F // S)
        per += per & 1
        for lo in range(0, F, per):
            n = min(per, F - lo)
            st = torch.cuda.Stream(ctx.device)
            with torch.cuda.stream(st):
                c = Context(0, st)
                fe = ImageFrontEnd(c, model, n, num_of_features=1000, kp_cap=1024, keep_panoramas=False)
                pipe = FramePairPipeline(c, rig, n // 2, frame_cap=8192, max_iter=10, seed=0, front_end=fe)
                fe.load_frames(omni3[lo:lo + n])
                keys2 = torch.zeros((n * model.nmask, fe.kp_cap, 2), dtype=torch.uint32, device=ctx.device)
            parts.append((st, c, fe, pipe, keys2))
        torch.cuda.synchronize()

        def c3_step():
            for st, c, fe, pipe, keys2 in parts:
                with torch.cuda.stream(st):
                    fe.run()
                    c.match_hamming(pipe.desc_bot, pipe.desc_top, pipe.n_bot, pipe.n_top, k=2, keys=keys2)  # the ratio rule's 2-NN
                    pipe.stereo()                                                                        # 1-NN keys, sort, gates, triangulation
        dt = timed(c3_step, lambda: torch.cuda.synchronize())
        n_view = np.mean([fe.n.cpu().numpy().reshape(2, -1, model.nmask).sum(-1).mean() for _, _, fe, _, _ in parts])
        M = np.mean([pipe.frames["M"].cpu().numpy().mean() for _, _, _, pipe, _ in parts])
        print(json.dumps({"config": "C3", "metric": "frames/s (unwrap + median + GFT + ORB descriptors + 2-NN bucket matching + triangulation), 1280x960 omni",
                          "value": F / dt, "ms_per_step": 1e3 * dt, "frames_per_step": F, "streams": len(parts),
                          "keypoints_per_view": float(n_view), "stereo_points_per_frame": float(M),
                          "data": "synthetic (%d distinct frames tiled)" % uniq}))
        for _, c, _, _, _ in parts:
            c.close()
        del parts
    # ---- C5
    if args.only == "C3":
        ctx.close()
        return
    cam = RGBDCamConfig(fx=554.256258, fy=554.256258, center_x=319.5, center_y=239.5, depth_is_Z=True, min_range=0.8, max_range=7.0)
    S5 = max(1, args.c5_streams)
    for algo in ("EPNP", "KNEIP"):
        # the batch as S5 one-call batches (sosvo_rgbd_pair_batch) on S5 HIP streams, one libsosvo context each
        insts = []
        for k in range(S5):
            st = torch.cuda.Stream(ctx.device)
            with torch.cuda.stream(st):
                c = Context(0, st)
                one = RGBDPairBatch(c, cam, B, num_of_features=2000, max_iter=2000, seed=1 + k * B, pose_est_algorithm=algo)
                one.load_frames(bgr, depth)
            insts.append((st, c, one))
        torch.cuda.synchronize()

        def c5_step():
            for st, c, one in insts:
                with torch.cuda.stream(st):
                    one.step()
        dt = timed(c5_step, lambda: torch.cuda.synchronize(), steps=5)
        rec = np.concatenate([one.results().cpu().numpy() for _, _, one in insts])
        print(json.dumps({"config": "C5", "metric": "frame-pairs/s (gray + GFT + ORB descriptors + back-projection + matching + central RANSAC 2000 it. + LM), 640x480 RGB-D",
                          "algorithm": algo, "value": S5 * B / dt, "ms_per_step": 1e3 * dt, "pairs_per_step": S5 * B, "streams": S5,
                          "tracked_ok": int((rec[:, 14] == 0).sum()), "inliers_mean": float(rec[:, 12].mean()),
                          "correspondences_mean": float(rec[:, 13].mean()), "data": "synthetic (16 distinct pairs tiled)"}))
        for _, c, _ in insts:
            c.close()
        del insts
    ctx.close()


if __name__ == "__main__":
    main()
