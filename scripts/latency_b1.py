"""Latency of ONE frame pair through the whole hot path (B = 1, one C-ABI call: sosvo_frame_pair_batch), host clock
around enqueue + synchronise; the per-frame figure a live VO loop sees.   python scripts/latency_b1.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    from vo_single_camera_sos_amd import synthetic
    from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
    from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=1440)
    gs.make_annulus_masks((480, 640))
    omni, _ = synthetic.make_frame_pairs(gs, 1, seed=3)
    from vo_single_camera_sos_amd.device import Context
    from vo_single_camera_sos_amd.frontend import DeviceImageModel
    from vo_single_camera_sos_amd.pipeline import FramePairBatch, RigConfig
    ctx = Context(0)
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    rig = RigConfig(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0], min_range=500.0,
                    max_range=7000.0, stereo_min_disp=1.0, stereo_max_hdiff=2.5, f2f_max_hdiff=0.125 * 0.5 * pano.cols,
                    pct_good_matches=1.0)
    model = DeviceImageModel(ctx, gs, (480, 640))
    out = {}
    for iters in (2000, 210):
        b = FramePairBatch(ctx, model, rig, 1, num_of_features=1000, kp_cap=512, frame_cap=2048, max_iter=iters, seed=1)
        b.load_frames(omni)
        for _ in range(5):
            b.step()
        ctx.synchronize()
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            b.step()
            ctx.synchronize()
            ts.append(time.perf_counter() - t0)
        rec = b.results().cpu().numpy()[0]
        out["ransac_%d" % iters] = {"median_ms": 1e3 * float(np.median(ts)), "min_ms": 1e3 * float(np.min(ts)),
                                    "inliers": int(rec[12]), "status": int(rec[14])}
    print(json.dumps({"config": "C2, one frame pair per call (B = 1)", "latency": out}))
    ctx.close()


if __name__ == "__main__":
    main()
