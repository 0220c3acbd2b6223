import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from vo_single_camera_sos_amd.device import Context
ctx = Context(0); dev = ctx.device
off = torch.from_numpy(np.stack([synth.F_TOP, synth.F_BOT])).to(dev); rot = torch.from_numpy(np.stack([np.eye(3)] * 2)).to(dev)
rng = np.random.default_rng(0)
pr = synth.make_abs_pose_problem(rng, 4096, inlier_frac=0.6, noise_deg=0.3, noncentral=True)
for B in (64, 8):
  for S in (4096, 1024):
    for npts in (64, 256, 780, min(3500, S)):
      for H in (2000, 500):
        f = np.zeros((B, S, 3)); p = np.zeros((B, S, 3)); cam = np.zeros((B, S), np.int32); n = np.full(B, npts, np.int32)
        f[:, :npts], p[:, :npts], cam[:, :npts] = pr["f"][:npts], pr["p"][:npts], np.sort(pr["cam"][:npts])
        tf, tp, tc, tn = [torch.from_numpy(a).to(dev) for a in (f, p, cam, n)]
        out = ctx.ransac_abs_pose(tf, tp, tn, synth.THR_5DEG, H, seed=1, cam=tc, cam_off=off, cam_rot=rot, cam_rot_identity=True)
        ctx.synchronize(); ctx.profile_enable(True)
        for rep in range(3):
            ctx.ransac_abs_pose(tf, tp, tn, synth.THR_5DEG, H, seed=1, cam=tc, cam_off=off, cam_rot=rot, cam_rot_identity=True, out=out)
        ctx.synchronize(); prof = ctx.profile_read(); ctx.profile_enable(False)
        t = min(ms for k, ms in prof if "score" in k)
        print("B=%2d stride=%4d n=%4d H=%4d  score %.3f ms" % (B, S, npts, H, t))
