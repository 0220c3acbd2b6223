"""Throughput of the BASELINE configurations that are parity-test cases rather than bench lines (bench.py measures C2):
  C3  1280x960 omni frames -> two 2880x292 panoramas, ~8000 keypoints per view: image front end (K1-K6), per-bucket
      2-NN Hamming matching + sort, stereo gates + bearings + midpoint triangulation (batched), frames/s
  C5  640x480 RGB-D frame pairs, ~2000 keypoints, central RANSAC (EPNP and KNEIP) + LM, one C-ABI call per step, pairs/s
Both split their batch over HIP streams (one libsosvo context each), as the C2 engine does.
One JSON line each; inputs resident in HBM, synthetic, every frame / pair DISTINCT (rendered by forked workers before the
process touches the GPU; --render-workers 1 under rocprofv3).  Each line carries its `roofline` (algorithmic
bytes per unit x units per launch of the dominant kernel / its average launch duration, HIP events recorded by the library
on its own streams; PMC traffic from the newest committed profiles/*/<c3|c5>*_pmc_hbm_per_kernel.csv) and the per-kernel
milliseconds per step.
    python scripts/bench_other_configs.py [--frames 192] [--pairs 128] [--only C3|C5]
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


_GS = None


def _render_c3(i):
    from vo_single_camera_sos_amd import synthetic
    return synthetic.render_omni(_GS, synthetic.Room(seed=70 + i, cells=(180.0, 45.0)), np.eye(3), np.zeros(3), 2.0,
                                 np.random.default_rng(70 + i))


def _render_c5(i):
    from vo_single_camera_sos_amd import synthetic
    rng = np.random.default_rng(900 + i)
    room = synthetic.Room(seed=900 + i, half_x=(1800.0, 2600.0), half_y=(2500.0, 3500.0), cells=(150.0, 40.0), yaw_deg=40.0)
    R, t = synthetic.random_step(rng, max_t=80.0, max_deg=8.0)
    ims, dps = [], []
    for (Rw, tw) in ((np.eye(3), np.zeros(3)), (R, t)):
        im, dp = synthetic.render_rgbd(room, Rw, tw, rng, depth_is_Z=True)
        ims.append(im)
        dps.append(dp)
    return ims, dps


def _pool_map(fn, items, workers):
    if workers > 1 and len(items) > 1:
        import multiprocessing
        with multiprocessing.get_context("fork").Pool(min(workers, len(items))) as pool:
            return pool.map(fn, items, chunksize=max(1, len(items) // (4 * workers)))
    return [fn(i) for i in items]


def kernel_table(prof, steps):
    """[(label, ms)] of the library's own HIP-event profile -> {label: [launches, total ms, min ms]}, dominant label."""
    per = {}
    for name, ms in prof:
        e = per.setdefault(name, [0, 0.0, float("inf")])
        e[0] += 1
        e[1] += ms
        e[2] = min(e[2], ms)
    # dominant by UNCONTENDED time (shortest launch x launches): the parts' launches overlap on the chip, so the summed
    # durations are shared time and would name whichever kernel waits most
    dom = max(per.items(), key=lambda kv: kv[1][2] * kv[1][0])
    return per, dom


def roofline(per, dom, b_alg_unit, units_per_launch, cfg_tag):
    """achieved = algorithmic bytes per launch of the dominant kernel / its average launch duration."""
    import csv
    import glob
    avg_s = dom[1][1] / dom[1][0] / 1e3
    min_s = dom[1][2] / 1e3
    achieved = b_alg_unit * units_per_launch / min_s / 1e9
    label = dom[0].strip("()").split("<")[0]
    traffic, src = None, None
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*%s*pmc_hbm_per_kernel.csv" % cfg_tag)))):
        for row in csv.DictReader(open(path)):
            if row["label"] == label and int(row["pairs_per_launch"] or 0) > 0:
                traffic = float(row["hbm_bytes_per_launch"]) * units_per_launch / float(row["pairs_per_launch"])
                src = os.path.relpath(path, ROOT)
                break
        if src:
            break
    return {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "frac_basis": "min_launch_ms (shortest launch in the timed region; dominant = largest min x launches)",
            "frac_avg_launch": b_alg_unit * units_per_launch / avg_s / 1e9 / 8000.0,
            "traffic": traffic, "traffic_source": src, "avg_launch_ms": avg_s * 1e3, "min_launch_ms": dom[1][2],
            "launches": dom[1][0], "algorithmic_bytes_per_unit": b_alg_unit, "units_per_launch": units_per_launch,
            "note": "VALU / latency-bound integer and FP64 work: the HBM fraction is small by construction (SURVEY 8d); "
                    "the launches of the batch's parts overlap on the chip, so avg_launch_ms includes shared time and "
                    "min_launch_ms is the kernel nearly on its own"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=192)
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--only", default="", help="C3 or C5: run just that configuration")
    ap.add_argument("--c3-streams", type=int, default=3, help="HIP streams the C3 batch is split over")
    ap.add_argument("--c5-streams", type=int, default=3, help="one-call RGB-D batches of --pairs pairs run side by side on this many HIP streams")
    ap.add_argument("--c3-pano-width", default="2400,2880",
                    help="C3 panorama columns, comma-separated: 2400 x 244 is SURVEY 8(d)'s C3 (the reference default scaled by two, "
                         "the geometry of tests/test_gpu_c3.py; GFT's minDistance and ORB.compute's border cap it at ~7400 keypoints "
                         "per view) -> config \"C3\"; 2880 x 292 reaches the ~8000+ keypoints per view the configuration names -> "
                         "config \"C3_2880\".  The same rendered omni frames serve every width")
    ap.add_argument("--render-workers", type=int, default=0, help="forked render processes (0 = auto; 1 under rocprofv3)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--c5-algo", default="both", choices=["both", "EPNP", "KNEIP"])
    ap.add_argument("--cache", default=None,
                    help=".npz the rendered inputs are kept in (the passes of scripts/profile_other.sh render once)")
    args = ap.parse_args()
    workers = args.render_workers if args.render_workers > 0 else max(1, min(16, len(os.sched_getaffinity(0))))
    from vo_single_camera_sos_amd import synthetic
    from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
    from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
    # ---- render on the host first (every C3 frame and every C5 pair its own room and noise)
    widths = [int(w) for w in str(args.c3_pano_width).split(",") if w]
    gs = synthetic_gums(scale=2.0)
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=widths[0])
    gs.make_annulus_masks((960, 1280))
    omni3 = None
    B = args.pairs
    S5 = max(1, args.c5_streams)
    bgr = depth = None
    cached = np.load(args.cache) if args.cache and os.path.exists(args.cache) else None
    if args.only != "C5":
        if cached is not None and "omni3" in cached and cached["omni3"].shape[0] == args.frames:
            omni3 = cached["omni3"]
        else:
            global _GS
            _GS = gs
            omni3 = np.stack(_pool_map(_render_c3, list(range(args.frames)), workers))
    if args.only != "C3":
        if cached is not None and "bgr" in cached and cached["bgr"].shape[0] == 2 * S5 * B:
            bgr, depth = cached["bgr"], cached["depth"]
        else:
            pairs = _pool_map(_render_c5, list(range(S5 * B)), workers)
            bgr = np.stack([im for p in pairs for im in p[0]])
            depth = np.stack([dp for p in pairs for dp in p[1]])
    if args.cache and cached is None:
        np.savez(args.cache, **{k: v for k, v in (("omni3", omni3), ("bgr", bgr), ("depth", depth)) if v is not None})

    import torch
    from vo_single_camera_sos_amd.device import Context
    from vo_single_camera_sos_amd.frontend import DeviceImageModel, ImageFrontEnd
    from vo_single_camera_sos_amd.pipeline import FramePairPipeline, RGBDCamConfig, RGBDPairBatch, RigConfig
    ctx = Context(0)
    for wi, width in enumerate(widths if args.only != "C5" else []):
        if wi > 0:
            gs = synthetic_gums(scale=2.0)
            for m in (gs.top_model, gs.bot_model):
                m.panorama = Panorama(m, width=width)
            gs.make_annulus_masks((960, 1280))
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
                if S > 1:
                    c.set_hint_shared_device(True)   # the parts run side by side (sosvo_set_hint)
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
        timed(c3_step, lambda: torch.cuda.synchronize(), warmup=2, steps=1)
        for _, c, _, _, _ in parts:
            c.profile_enable(True)
        dt = timed(c3_step, lambda: torch.cuda.synchronize(), warmup=0, steps=args.steps)
        prof = []
        for _, c, _, _, _ in parts:
            prof.extend(c.profile_read())
            c.profile_enable(False)
        per, dom = kernel_table(prof, args.steps)
        n_view = np.mean([fe.n.cpu().numpy().reshape(2, -1, model.nmask).sum(-1).mean() for _, _, fe, _, _ in parts])
        M = np.mean([pipe.frames["M"].cpu().numpy().mean() for _, _, _, pipe, _ in parts])
        cap_hit = any(int(fe.n.max().item()) >= fe.kp_cap for _, _, fe, _, _ in parts)
        N = int(round(float(n_view)))
        b_alg = 960 * 1280 * 3 + N * 12 + N * 24   # one BGR frame in, match index + distance and XYZ out
        print(json.dumps({"config": "C3" if width == 2400 else "C3_%d" % width, "metric": "frames/s (unwrap + median + GFT + ORB descriptors + 2-NN bucket matching + triangulation), 1280x960 omni",
                          "value": F / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt, "frames_per_step": F, "streams": len(parts),
                          "panorama": "%d x %d" % (pano.cols, pano.rows), "keypoints_per_view": float(n_view),
                          "keypoint_capacity_hit": bool(cap_hit), "stereo_points_per_frame": float(M),
                          "roofline": roofline(per, dom, b_alg, F // len(parts), "c3" if width == 2880 else "c3w%d" % width),
                          "kernels_ms_per_step": {k: v[1] / args.steps for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])},
                          "data": "synthetic (%d distinct frames)" % F}))
        for _, c, _, _, _ in parts:
            c.close()
        del parts
    # ---- C5
    if args.only == "C3":
        ctx.close()
        return
    cam = RGBDCamConfig(fx=554.256258, fy=554.256258, center_x=319.5, center_y=239.5, depth_is_Z=True, min_range=0.8, max_range=7.0)
    for algo in (("EPNP", "KNEIP") if args.c5_algo == "both" else (args.c5_algo,)):
        # the batch as S5 one-call batches (sosvo_rgbd_pair_batch) on S5 HIP streams, one libsosvo context each
        insts = []
        for k in range(S5):
            st = torch.cuda.Stream(ctx.device)
            with torch.cuda.stream(st):
                c = Context(0, st)
                one = RGBDPairBatch(c, cam, B, num_of_features=2000, max_iter=2000, seed=1 + k * B, pose_est_algorithm=algo)
                one.load_frames(bgr[2 * k * B:2 * (k + 1) * B], depth[2 * k * B:2 * (k + 1) * B])
            insts.append((st, c, one))
        torch.cuda.synchronize()

        def c5_step():
            for st, c, one in insts:
                with torch.cuda.stream(st):
                    one.step()
        timed(c5_step, lambda: torch.cuda.synchronize(), warmup=2, steps=1)
        for _, c, _ in insts:
            c.profile_enable(True)
        dt = timed(c5_step, lambda: torch.cuda.synchronize(), warmup=0, steps=max(2, args.steps // 2))
        prof = []
        for _, c, _ in insts:
            prof.extend(c.profile_read())
            c.profile_enable(False)
        nsteps = max(2, args.steps // 2)
        per, dom = kernel_table(prof, nsteps)
        rec = np.concatenate([one.results().cpu().numpy() for _, _, one in insts])
        N = 2000
        b_alg = 2 * (480 * 640 * 3 + 480 * 640 * 2) + N * 12 + N + 96   # two BGR + u16 depth frames in; matches, mask, pose out
        print(json.dumps({"config": "C5", "metric": "frame-pairs/s (gray + GFT + ORB descriptors + back-projection + matching + central RANSAC 2000 it. + LM), 640x480 RGB-D",
                          "algorithm": algo, "value": S5 * B / dt, "unit": "frame-pairs/s", "ms_per_step": 1e3 * dt, "pairs_per_step": S5 * B, "streams": S5,
                          "tracked_ok": int((rec[:, 14] == 0).sum()), "inliers_mean": float(rec[:, 12].mean()),
                          "correspondences_mean": float(rec[:, 13].mean()),
                          "roofline": roofline(per, dom, b_alg, B, "c5" + algo.lower()),
                          "kernels_ms_per_step": {k: v[1] / nsteps for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])},
                          "data": "synthetic (%d distinct pairs)" % (S5 * B)}))
        for _, c, _ in insts:
            c.close()
        del insts
    ctx.close()


if __name__ == "__main__":
    main()
