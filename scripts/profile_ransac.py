"""Isolated per-kernel times of sosvo_ransac_abs_pose on synthetic non-central problems (one stream, nothing else on the
chip):   python scripts/profile_ransac.py [--problems 256] [--points 770] [--iters 2000] [--solver GP3P|P3P]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=256)
    ap.add_argument("--points", type=int, default=770)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--solver", default="GP3P")
    args = ap.parse_args()
    import torch
    import synth
    from vo_single_camera_sos_amd.device import Context
    rng = np.random.default_rng(3)
    base = [synth.make_abs_pose_problem(rng, args.points, inlier_frac=0.84, noise_deg=0.3, noncentral=True) for _ in range(8)]
    P, S = args.problems, 1024
    f = np.zeros((P, S, 3)); p = np.zeros((P, S, 3)); cam = np.zeros((P, S), np.int32); n = np.zeros(P, np.int32)
    for b in range(P):
        pr = base[b % 8]
        k = pr["f"].shape[0]
        n[b] = k; f[b, :k] = pr["f"]; p[b, :k] = pr["p"]; cam[b, :k] = pr["cam"]
    ctx = Context(0)
    dev = ctx.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    tf, tp, tc, tn = t(f), t(p), t(cam), t(n)
    off, rot = t(base[0]["cam_off"]), t(base[0]["cam_rot"])
    kw = dict(cam=tc, cam_off=off, cam_rot=rot, cam_rot_identity=True, gp3p=args.solver == "GP3P")
    for _ in range(2):
        ctx.ransac_abs_pose(tf, tp, tn, synth.THR_5DEG, args.iters, seed=1, **kw)
    ctx.synchronize()
    ctx.profile_enable(True)
    steps = 5
    for _ in range(steps):
        out = ctx.ransac_abs_pose(tf, tp, tn, synth.THR_5DEG, args.iters, seed=1, **kw)
    ctx.synchronize()
    acc = collections.OrderedDict()
    for name, ms in ctx.profile_read():
        acc[name] = acc.get(name, 0.0) + ms / steps
    for name, ms in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("%-44s %8.3f ms" % (name, ms))
    print("valid hypotheses per problem:", out["info"][:4, 3].tolist(), "inliers", out["n_inliers"][:4].tolist())


if __name__ == "__main__":
    main()
