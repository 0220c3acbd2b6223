"""GPU parity: sosvo_ransac_rel_pose (2D-2D relative-pose RANSAC, through the C ABI) against the CPU oracle on the same
seeded inputs: per-hypothesis inlier counts, iterations drawn, winning iteration, inlier mask / ascending index list
(integer work: bit-exact) and the pose (same IEEE operations in the same order on both sides: bit-exact too)."""
import numpy as np
import pytest
import torch

import oracle
from test_oracle_relpose import _two_views

pytestmark = pytest.mark.gpu

THR = 2.0 * (1.0 - np.cos(np.deg2rad(1.0)))


def _compare(ctx, problems, S, max_iter, seed, adaptive, algorithm=8):
    P = len(problems)
    f1 = np.zeros((P, S, 3))
    f2 = np.zeros((P, S, 3))
    n = np.zeros(P, dtype=np.int32)
    for b, (a, c) in enumerate(problems):
        n[b] = a.shape[0]
        f1[b, :n[b]] = a
        f2[b, :n[b]] = c
    dev = ctx.device
    out = ctx.ransac_rel_pose(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev), torch.from_numpy(n).to(dev), THR,
                              max_iter, algorithm=algorithm, seed=seed, adaptive=adaptive, want_counts=True)
    ctx.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    for b, (a, c) in enumerate(problems):
        want = oracle.ransac_rel_pose(a, c, THR, max_iter, seed=seed + b, adaptive=adaptive, want_counts=True,
                                      algorithm=algorithm)
        used = want["iters_used"]
        assert got["info"][b, 1] == used, "iterations drawn, problem %d" % b
        assert np.array_equal(got["counts"][b, :used], want["counts"][:used]), "hypothesis counts, problem %d" % b
        assert got["info"][b, 0] == want["best_iter"] and got["info"][b, 2] == want["status"]
        assert np.array_equal(got["mask"][b, :n[b]].astype(bool), want["mask"])
        k = want["n_inliers"]
        assert got["n_inliers"][b] == k
        assert np.array_equal(got["idx"][b, :k], np.flatnonzero(want["mask"]))
        assert np.array_equal(got["T"][b].view(np.uint64), want["T"].view(np.uint64)), "pose bits, problem %d" % b
    return got


def test_eightpt_ransac_matches_the_oracle_bit_for_bit(ctx):
    rng = np.random.default_rng(11)
    problems = []
    for n, noise, outl in ((500, 0.05, 0.3), (64, 0.0, 0.0), (257, 0.2, 0.5), (8, 0.0, 0.0), (7, 0.0, 0.0), (1000, 0.1, 0.2)):
        a, c, _, _, _ = _two_views(rng, n, noise_deg=noise, outlier_frac=outl)
        problems.append((a, c))
    got = _compare(ctx, problems, 1024, 300, seed=3, adaptive=False)
    assert got["info"][4, 2] == 1 and got["n_inliers"][4] == 0          # fewer than 8 correspondences: no model
    assert np.array_equal(got["T"][4], np.eye(3, 4))
    _compare(ctx, problems, 1024, 2000, seed=9, adaptive=True)


@pytest.mark.parametrize("algorithm,need", [(5, 8), (7, 9)])
def test_fivept_and_sevenpt_ransac_match_the_oracle_bit_for_bit(ctx, algorithm, need):
    rng = np.random.default_rng(13 + algorithm)
    problems = []
    for n, noise, outl in ((400, 0.05, 0.3), (need, 0.0, 0.0), (need - 1, 0.0, 0.0), (130, 0.1, 0.1)):
        a, c, _, _, _ = _two_views(rng, n, noise_deg=noise, outlier_frac=outl)
        problems.append((a, c))
    got = _compare(ctx, problems, 512, 200, seed=5, adaptive=False, algorithm=algorithm)
    assert got["info"][2, 2] == 1 and got["info"][1, 2] == 0
    assert (got["counts"][0] >= 0).mean() > 0.95
    _compare(ctx, problems, 512, 1000, seed=6, adaptive=True, algorithm=algorithm)


def test_pyopengv_face_and_the_reference_wrapper(ctx):
    from vo_single_camera_sos_amd import pyopengv
    from vo_single_camera_sos_amd.omnistereo import pose_est_tools
    rng = np.random.default_rng(12)
    f1, f2, R, t, good = _two_views(rng, 400, noise_deg=0.05, outlier_frac=0.3)
    pyopengv.set_seed(21)
    T, inl = pyopengv.relative_pose_ransac(f1, f2, "EIGHTPT", THR, 500)
    want = oracle.ransac_rel_pose(f1, f2, THR, 500, seed=21, adaptive=True)
    assert np.array_equal(T, want["T"]) and np.array_equal(inl, np.flatnonzero(want["mask"]))
    ang = np.arccos(np.clip((np.trace(T[:, :3].T @ R) - 1) / 2, -1, 1))
    assert ang < np.deg2rad(1.0) and len(inl) >= 0.9 * good.sum()
    T4, inl4 = pose_est_tools.pose_relative_ransac_2D_to_2D(np.hstack([f1, np.ones((400, 1))]), np.hstack([f2, np.ones((400, 1))]),
                                                            model_error_threshold=THR, rel_pose_est_algorithm="EIGHTPT")
    assert T4.shape == (4, 4) and np.array_equal(T4[3], [0, 0, 0, 1]) and len(inl4) >= 0.9 * good.sum()
    # the reference's default algorithm (pose_est_tools.py:54: "STEWENIUS", budget from w = 0.5 and 5 points: 239 iterations)
    pyopengv.set_seed(33)
    T5, inl5 = pose_est_tools.pose_relative_ransac_2D_to_2D(f1, f2, model_error_threshold=THR)
    want5 = oracle.ransac_rel_pose(f1, f2, THR, 239, seed=33, adaptive=True, algorithm=5)
    assert np.array_equal(T5[:3], want5["T"]) and np.array_equal(inl5, np.flatnonzero(want5["mask"]))
    assert np.arccos(np.clip((np.trace(T5[:3, :3].T @ R) - 1) / 2, -1, 1)) < np.deg2rad(1.0)
    with pytest.raises(ValueError):
        pyopengv.relative_pose_ransac(f1, f2, "NOPE", THR, 10)


@pytest.mark.parametrize("algorithm", [5, 7, 8])
def test_degenerate_inputs_agree_with_the_oracle(ctx, algorithm):
    """No motion at all (f2 = f1: the essential matrix is undefined), all bearings equal, points on a plane seen under a
    pure rotation: whatever the solvers make of it, the device makes the same of it, bit for bit, and nothing hangs."""
    rng = np.random.default_rng(40 + algorithm)
    a, c, R, t, _ = _two_views(rng, 60)
    same = (a.copy(), a.copy())
    one = (np.tile(a[:1], (40, 1)), np.tile(a[:1], (40, 1)))
    rot_only = (a.copy(), np.ascontiguousarray(a @ R))                 # X2 = R^T X1: zero baseline
    got = _compare(ctx, [same, one, rot_only, (a, c)], 64, 120, seed=77, adaptive=False, algorithm=algorithm)
    assert np.isfinite(got["T"]).all()
    assert got["info"][3, 2] == 0 and got["n_inliers"][3] >= 50            # the proper problem next to them is solved
