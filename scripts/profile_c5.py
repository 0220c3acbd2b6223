"""Per-kernel milliseconds of one step of config 5 (RGB-D pairs through sosvo_rgbd_pair_batch), from the library's own
HIP-event profile:   python scripts/profile_c5.py [--pairs 128] [--algo EPNP]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--algo", default="EPNP")
    args = ap.parse_args()
    from vo_single_camera_sos_amd import synthetic
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
    from vo_single_camera_sos_amd.device import Context
    from vo_single_camera_sos_amd.pipeline import RGBDCamConfig, RGBDPairBatch
    ctx = Context(0)
    cam = RGBDCamConfig(fx=554.256258, fy=554.256258, center_x=319.5, center_y=239.5, depth_is_Z=True, min_range=0.8, max_range=7.0)
    one = RGBDPairBatch(ctx, cam, B, num_of_features=2000, max_iter=2000, seed=1, pose_est_algorithm=args.algo)
    one.load_frames(bgr, depth)
    for _ in range(2):
        one.step()
    ctx.synchronize()
    ctx.profile_enable(True)
    steps = 5
    for _ in range(steps):
        one.step()
    ctx.synchronize()
    acc = collections.OrderedDict()
    for name, ms in ctx.profile_read():
        acc[name] = acc.get(name, 0.0) + ms / steps
    tot = sum(acc.values())
    for name, ms in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("%-44s %8.3f ms" % (name, ms))
    print("total %.3f ms per step of %d pairs -> %.0f pairs/s if back to back" % (tot, B, B / tot * 1e3))


if __name__ == "__main__":
    main()
