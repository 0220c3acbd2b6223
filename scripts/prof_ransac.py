"""Ad-hoc: per-kernel times of sosvo_ransac_abs_pose for C2-like batches (GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from vo_single_camera_sos_amd.device import Context
ctx = Context(0)
dev = ctx.device
off = torch.from_numpy(np.stack([synth.F_TOP, synth.F_BOT])).to(dev); rot = torch.from_numpy(np.stack([np.eye(3)] * 2)).to(dev)
for npts in (780, 3500):
    rng = np.random.default_rng(0)
    B, S = 64, 4096
    f = np.zeros((B, S, 3)); p = np.zeros((B, S, 3)); cam = np.zeros((B, S), np.int32); n = np.full(B, npts, np.int32)
    pr = synth.make_abs_pose_problem(rng, npts, inlier_frac=0.6, noise_deg=0.3, noncentral=True)
    f[:, :npts], p[:, :npts], cam[:, :npts] = pr["f"], pr["p"], pr["cam"]
    tf, tp, tc, tn = [torch.from_numpy(a).to(dev) for a in (f, p, cam, n)]
    out = ctx.ransac_abs_pose(tf, tp, tn, synth.THR_5DEG, 2000, seed=1, cam=tc, cam_off=off, cam_rot=rot, cam_rot_identity=True)
    ctx.synchronize()
    ctx.profile_enable(True)
    for rep in range(5):
        ctx.ransac_abs_pose(tf, tp, tn, synth.THR_5DEG, 2000, seed=1, cam=tc, cam_off=off, cam_rot=rot, cam_rot_identity=True, out=out)
    ctx.synchronize()
    prof = ctx.profile_read(); ctx.profile_enable(False)
    agg = {}
    for k, ms in prof: agg.setdefault(k, []).append(ms)
    scores = 64 * 2000 * npts
    for k, v in agg.items():
        extra = "  -> %.1f G scores/s" % (scores / (min(v) * 1e-3) / 1e9) if "score" in k else ""
        print("n=%d %-36s %.3f ms%s" % (npts, k, min(v), extra))
    print("  inliers", out["n_inliers"][:3].tolist())
