"""A/B of the scoring kernel's forms on one box (mode 0 = sosvo_set_hint(SOSVO_HINT_SCORE_FP64_ONLY): double precision
only, 1 = the default, single-precision tier in front; lib=<path> = another build of the library): isolated time of ransac_score_kernel on C2-shaped problems (P3P hypotheses, identity camera rotations) and
a hash of the per-hypothesis inlier counts, which every form must reproduce bit for bit.

    python scripts/score_tiers.py [--problems 256] [--points 1285] [--iters 2000] [--modes 0,1] [--no-ident]"""
import argparse
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(args):
    import collections
    import numpy as np
    import torch
    import synth
    if os.environ.get("SOSVO_AB_LIB"):  # another build of the library (A/B against an earlier commit)
        from vo_single_camera_sos_amd import _lib
        _lib.LIB_PATH = os.environ["SOSVO_AB_LIB"]
    from vo_single_camera_sos_amd.device import Context
    rng = np.random.default_rng(3)
    base = [synth.make_abs_pose_problem(rng, args.points - 7 * k, inlier_frac=0.9, noise_deg=0.3, noncentral=True) for k in range(8)]
    P, S = args.problems, -(-args.points // 256) * 256
    f = np.zeros((P, S, 3)); p = np.zeros((P, S, 3)); cam = np.zeros((P, S), np.int32); n = np.zeros(P, np.int32)
    for b in range(P):
        pr = base[b % 8]
        k = pr["f"].shape[0]
        n[b] = k; f[b, :k] = pr["f"]; p[b, :k] = pr["p"]; cam[b, :k] = pr["cam"]
    ctx = Context(0)
    if os.environ.get("SOSVO_SCORE_T1") == "0" and not os.environ.get("SOSVO_AB_LIB"):
        ctx.set_hint_score_fp64_only(True)
    dev = ctx.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    tf, tp, tc, tn = t(f), t(p), t(cam), t(n)
    off, rot = t(base[0]["cam_off"]), t(base[0]["cam_rot"])
    kw = dict(cam=tc, cam_off=off, cam_rot=rot, cam_rot_identity=not args.no_ident, gp3p=args.solver == "GP3P", want_counts=True)
    for _ in range(2):
        out = ctx.ransac_abs_pose(tf, tp, tn, synth.THR_5DEG, args.iters, seed=1, **kw)
    ctx.synchronize()
    ctx.profile_enable(True)
    steps = 5
    for _ in range(steps):
        out = ctx.ransac_abs_pose(tf, tp, tn, synth.THR_5DEG, args.iters, seed=1, out=out, **kw)
    ctx.synchronize()
    acc = collections.OrderedDict()
    for name, ms in ctx.profile_read():
        acc[name] = acc.get(name, 0.0) + ms / steps
    score = sum(ms for name, ms in acc.items() if "ransac_score" in name)
    h = hashlib.sha256()
    for k in ("counts", "n_inliers", "info", "T", "mask"):
        h.update(out[k].cpu().numpy().tobytes())
    print("lib %s mode %s: ransac_score_kernel %.3f ms per %d problems x %d points x %d hypotheses; inliers %s; sha %s" % (
        os.path.basename(os.environ.get("SOSVO_AB_LIB", "libsosvo.so")), os.environ.get("SOSVO_SCORE_T1", "default"), score, P, args.points, args.iters, out["n_inliers"][:3].tolist(),
        h.hexdigest()[:16]), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=256)
    ap.add_argument("--points", type=int, default=1285)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--solver", default="P3P")
    ap.add_argument("--modes", default="0,1")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--no-ident", action="store_true", help="camera rotations treated as general matrices (the PPT = 1 form)")
    args = ap.parse_args()
    if args.child:
        return child(args)
    rc = 0
    for m in args.modes.split(","):
        env = dict(os.environ, SOSVO_SCORE_T1=m)
        if m.startswith("lib="):  # lib=<path>: that build, its own default form
            env = dict(os.environ, SOSVO_AB_LIB=os.path.join(ROOT, m[4:]))
        rc |= subprocess.call([sys.executable, os.path.abspath(__file__), "--child", "--problems", str(args.problems), "--points",
                               str(args.points), "--iters", str(args.iters), "--solver", args.solver] +
                              (["--no-ident"] if args.no_ident else []), env=env)
    sys.exit(rc)


if __name__ == "__main__":
    main()
