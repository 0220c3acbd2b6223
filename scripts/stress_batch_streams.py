"""Development aid: many steps of sosvo_frame_pair_batch_streams (two internal streams) against the one-stream call;
prints the steps whose records differ (none expected).  STEPS=n, POISON=byte (fill the workspace before every step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_batch_call import _setup
from vo_single_camera_sos_amd.device import Context
from vo_single_camera_sos_amd.pipeline import FramePairBatch
B, nfeat, cap = 4, 300, 320
ctx0 = Context(0)
model, rig, omni = _setup(ctx0, B)
omni[3] = 0
one = FramePairBatch(ctx0, model, rig, B, num_of_features=nfeat, kp_cap=cap, frame_cap=1024, max_iter=300, seed=11)
one.load_frames(omni)
want = one.step().clone(); ctx0.synchronize(); want = want.cpu().numpy()
ctx = Context(0)
multi = FramePairBatch(ctx, model, rig, B, num_of_features=nfeat, kp_cap=cap, frame_cap=1024, max_iter=300, seed=11, n_streams=2)
multi.load_frames(omni)
bad_steps = []
N = int(os.environ.get("STEPS", "400"))
for it in range(N):
    if os.environ.get("POISON"):
        multi.workspace.fill_(int(os.environ["POISON"]))
    g = multi.step().clone(); ctx.synchronize(); g = g.cpu().numpy()
    if not np.array_equal(g, want):
        bad_steps.append(it)
print("mismatching steps of %d:" % N, bad_steps[:20], len(bad_steps))
