"""Ad-hoc: time the LM refinement kernel for different iteration caps (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, synth
from vo_single_camera_sos_amd.device import Context
ctx = Context(0)
rng = np.random.default_rng(0)
B, S = 64, 1024
f = np.zeros((B, S, 3)); p = np.zeros((B, S, 3)); cam = np.zeros((B, S), np.int32); n = np.zeros(B, np.int32); T0 = np.zeros((B, 3, 4))
for b in range(B):
    pr = synth.make_abs_pose_problem(rng, 780, inlier_frac=0.85, noise_deg=0.3, noncentral=True)
    f[b, :780], p[b, :780], cam[b, :780], n[b] = pr["f"], pr["p"], pr["cam"], 780
    T0[b] = np.hstack([pr["R"] @ synth.rot_from_axis_angle([1, 2, 3], 0.01), (pr["t"] + [3., -2, 1])[:, None]])
dev = ctx.device
tf, tp, tc, tn = [torch.from_numpy(a).to(dev) for a in (f, p, cam, n)]
off = torch.from_numpy(np.stack([synth.F_TOP, synth.F_BOT])).to(dev); rot = torch.from_numpy(np.stack([np.eye(3)] * 2)).to(dev)
for iters in (1, 2, 4, 8, 16, 30):
    ts = []
    for rep in range(5):
        T = torch.from_numpy(T0).to(dev)
        ctx.synchronize(); ctx.timer_start()
        _, cost, its = ctx.refine_abs_pose(tf, tp, tn, T, cam=tc, cam_off=off, cam_rot=rot, max_lm_iter=iters)
        ctx.timer_stop(); ts.append(ctx.timer_elapsed_ms())
    print("max_lm_iter %2d: %.3f ms  (iters used: mean %.1f max %d)" % (iters, min(ts), its.float().mean().item(), its.max().item()))
