#!/usr/bin/env python3
"""Outside evidence for the (unpinned) absolute-pose solvers on the SOS path: pose error of the batched engine against
the PLANTED motion of the synthetic pairs, for the reference's 5-degree RANSAC threshold (pose_est_tools.py:675-676) and
tighter ones, for the one-mirror P3P and the generalised P3P generators, and by range bin of the triangulated points.

    python scripts/accuracy_sos.py [--pairs 256] [--json out.json]

Run on the GPU box.  Prints one JSON document."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def errors(rec, poses):
    rot, tra = [], []
    for i, (R, t) in enumerate(poses):
        T = rec[i, :12].reshape(3, 4)
        dR = T[:, :3].T @ R
        rot.append(np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))))
        tra.append(np.linalg.norm(T[:, 3] - t))   # mm
    return np.array(rot), np.array(tra)


def summary(rot, tra, ok):
    q = lambda a, p: float(np.percentile(a[ok], p)) if ok.any() else None  # noqa: E731
    return {"rot_deg_median": q(rot, 50), "rot_deg_p90": q(rot, 90), "rot_deg_max": q(rot, 100),
            "trans_mm_median": q(tra, 50), "trans_mm_p90": q(tra, 90), "trans_mm_max": q(tra, 100), "tracked_ok": int(ok.sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--pano-width", type=int, default=1440)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from vo_single_camera_sos_amd import synthetic
    from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
    from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
    from vo_single_camera_sos_amd.pipeline import OverlappedFramePairs, RigConfig
    H, W, B = 480, 640, args.pairs
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=args.pano_width)
    gs.make_annulus_masks((H, W))
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    omni, poses = synthetic.make_frame_pairs(gs, B, seed=args.seed, workers=min(16, os.cpu_count() or 1))
    out = {"pairs": B, "panorama": "%d x %d" % (pano.cols, pano.rows), "iterations": args.iters, "runs": []}
    for max_range in (7000.0, 3000.0):
        rig_kw = dict(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0], min_range=500.0,
                      max_range=max_range, stereo_min_disp=1.0, stereo_max_hdiff=2.5, f2f_max_hdiff=0.125 * 0.5 * pano.cols,
                      pct_good_matches=1.0)
        for solver in ("P3P", "GP3P"):
            for thr_deg in (5.0, 2.0, 1.0, 0.5):
                eng = OverlappedFramePairs(0, gs, (H, W), RigConfig(**rig_kw), B, n_streams=2, num_of_features=1000, kp_cap=512,
                                           frame_cap=2048, max_iter=args.iters, seed=args.seed, ransac_solver=solver,
                                           thr=float(1.0 - np.cos(np.deg2rad(thr_deg))))
                eng.load_frames(omni)
                eng.step()
                rec = eng.results().cpu().numpy()
                # mean range of the reference frame's triangulated points of each pair (frame 2i of its part)
                rng_mean = []
                for p in eng.parts:
                    X = p.pipe.frames["X"].cpu().numpy()
                    M = p.pipe.frames["M"].cpu().numpy()
                    for f in range(0, X.shape[0], 2):
                        rng_mean.append(float(np.linalg.norm(X[f, :M[f]], axis=1).mean()) if M[f] else float("nan"))
                eng.close()
                rot, tra = errors(rec, poses)
                ok = rec[:, 14] == 0
                run = {"max_range_mm": max_range, "solver": solver, "threshold_deg": thr_deg,
                       "inliers_mean": float(rec[:, 12].mean()), "correspondences_mean": float(rec[:, 13].mean())}
                run.update(summary(rot, tra, ok))
                rng_mean = np.array(rng_mean)
                bins = np.nanpercentile(rng_mean, [0, 33, 67, 100])
                run["by_mean_point_range_mm"] = []
                for lo, hi in zip(bins[:-1], bins[1:]):
                    sel = ok & (rng_mean >= lo) & (rng_mean <= hi)
                    run["by_mean_point_range_mm"].append({"lo": float(lo), "hi": float(hi), "pairs": int(sel.sum()),
                                                          "rot_deg_median": float(np.median(rot[sel])) if sel.any() else None,
                                                          "trans_mm_median": float(np.median(tra[sel])) if sel.any() else None})
                out["runs"].append(run)
                print("max_range %.0f %s thr %.1f: rot median %.3f p90 %.3f max %.3f deg, trans median %.2f p90 %.2f max %.2f mm, "
                      "inliers %.0f / %.0f" % (max_range, solver, thr_deg, run["rot_deg_median"], run["rot_deg_p90"], run["rot_deg_max"],
                                               run["trans_mm_median"], run["trans_mm_p90"], run["trans_mm_max"],
                                               run["inliers_mean"], run["correspondences_mean"]), file=sys.stderr)
    s = json.dumps(out)
    print(s)
    if args.json:
        open(args.json, "w").write(s + "\n")


if __name__ == "__main__":
    main()
