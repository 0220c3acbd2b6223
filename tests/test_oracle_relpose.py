"""2D-2D relative-pose RANSAC restatement (oracle/relpose_core.h): the SCORE is pinned by the reference's own code
(tests/golden/geometry_c2.npz does not hold it, so it is re-derived here in numpy from pose_est_tools.py:150-203); the
eight-, seven- and five-point solvers are checked against planted motions and independent numpy evaluations (SVD null
space, epipolar residuals, the essential-matrix constraints); RANSAC finds the planted inliers with each of them."""
import numpy as np

import oracle
import synth


def _two_views(rng, n, noise_deg=0.0, outlier_frac=0.0):
    R, t = synth.random_pose(rng, max_t=400.0, max_deg=20.0)
    X1 = rng.normal(size=(n, 3))
    X1 = X1 / np.linalg.norm(X1, axis=1, keepdims=True) * rng.uniform(1500.0, 6000.0, (n, 1))
    X2 = (X1 - t) @ R                                    # X1 = R X2 + t
    f1 = X1 / np.linalg.norm(X1, axis=1, keepdims=True)
    f2 = synth.perturb_bearings(rng, X2 / np.linalg.norm(X2, axis=1, keepdims=True), noise_deg)
    bad = rng.random(n) < outlier_frac
    g = rng.normal(size=(n, 3))
    f2 = np.where(bad[:, None], g / np.linalg.norm(g, axis=1, keepdims=True), f2)
    return np.ascontiguousarray(f1), np.ascontiguousarray(f2), R, t, ~bad


def _score_numpy(T, f1, f2):
    """pose_est_tools.py:150-203, relative case, written out in numpy (triangulate2 as SURVEY App. E.8)."""
    R, t = T[:, :3], T[:, 3]
    g = R @ f2
    A = np.array([[f1 @ f1, -(f1 @ g)], [f1 @ g, -(g @ g)]])
    lam = np.linalg.solve(A, np.array([t @ f1, t @ g]))
    X = (lam[0] * f1 + t + lam[1] * g) / 2.0
    x2 = R.T @ (X - t)
    return (1.0 - f1 @ X / np.linalg.norm(X)) + (1.0 - f2 @ x2 / np.linalg.norm(x2))


def test_score_is_the_references_definition():
    rng = np.random.default_rng(1)
    f1, f2, R, t, _ = _two_views(rng, 50, noise_deg=0.5)
    T = np.concatenate([R, (t / np.linalg.norm(t))[:, None]], axis=1)
    for i in range(50):
        assert abs(oracle.rel_score(T, f1[i], f2[i]) - _score_numpy(T, f1[i], f2[i])) < 1e-13
    f1c, f2c, R, t, _ = _two_views(rng, 10)
    T = np.concatenate([R, (t / np.linalg.norm(t))[:, None]], axis=1)
    assert max(oracle.rel_score(T, a, b) for a, b in zip(f1c, f2c)) < 1e-12      # the true motion scores zero


def test_eightpt_recovers_planted_motions_and_spans_the_null_space():
    rng = np.random.default_rng(2)
    worst = 0.0
    for _ in range(100):
        f1, f2, R, t, _ = _two_views(rng, 8)
        T = oracle.eightpt(f1, f2)
        assert T is not None
        tn = t / np.linalg.norm(t)
        worst = max(worst, np.abs(T[:, :3] - R).max(), np.abs(T[:, 3] - tn).max())
        # independent: E = [t]x R from the result annihilates the correspondences, and numpy's null vector of A is parallel
        tx = np.array([[0, -T[2, 3], T[1, 3]], [T[2, 3], 0, -T[0, 3]], [-T[1, 3], T[0, 3], 0]])
        E = tx @ T[:, :3]
        assert max(abs(a @ E @ b) for a, b in zip(f1, f2)) < 1e-9
        A = np.stack([np.outer(a, b).ravel() for a, b in zip(f1, f2)])
        e = np.linalg.svd(A)[2][-1]
        assert abs(abs(e @ E.ravel()) / np.linalg.norm(E) - 1.0) < 1e-8
    assert worst < 1e-7, worst


def _true_essential(R, t):
    tn = t / np.linalg.norm(t)
    tx = np.array([[0, -tn[2], tn[1]], [tn[2], 0, -tn[0]], [-tn[1], tn[0], 0]])
    E = tx @ R
    return E / np.linalg.norm(E)


def _nearest(Es, Et):
    return min((min(np.linalg.norm(e / np.linalg.norm(e) - Et), np.linalg.norm(e / np.linalg.norm(e) + Et)) for e in Es),
               default=9.0)


def test_fivept_solutions_are_essential_matrices_of_the_sample_and_contain_the_planted_motion():
    """Every matrix returned (i) annihilates the five correspondences, (ii) lies in numpy's SVD null space of the
    constraint matrix, (iii) satisfies det E = 0 and 2 E E^T E = tr(E E^T) E -- i.e. it IS a solution of the five-point
    problem, checked without the solver's code; and the planted motion is among the <= 10 solutions (a fraction of a
    percent of random samples is lost to close root pairs of the degree-ten polynomial: the bar is 99 %)."""
    rng = np.random.default_rng(4)
    found, counts = 0, []
    for _ in range(400):
        f1, f2, R, t, _ = _two_views(rng, 5)
        Es = oracle.fivept(f1, f2)
        assert 1 <= len(Es) <= 10
        counts.append(len(Es))
        A = np.stack([np.outer(a, b).ravel() for a, b in zip(f1, f2)])
        null = np.linalg.svd(A)[2][5:]                                   # orthonormal rows spanning the null space
        for E in Es:
            assert max(abs(a @ E @ b) for a, b in zip(f1, f2)) < 1e-10
            assert np.linalg.norm(E.ravel() - null.T @ (null @ E.ravel())) < 1e-10 * np.linalg.norm(E)
            En = E / np.linalg.norm(E)
            assert abs(np.linalg.det(En)) < 1e-9 and np.abs(2 * En @ En.T @ En - np.trace(En @ En.T) * En).max() < 2e-9
            sv = np.linalg.svd(En, compute_uv=False)
            assert abs(sv[0] - sv[1]) < 1e-8 and sv[2] < 1e-8            # two equal singular values and a zero
        found += _nearest(Es, _true_essential(R, t)) < 1e-8
    assert found >= 396, found
    assert max(counts) >= 8 and min(counts) >= 2                          # real-solution counts vary up to ten


def test_sevenpt_returns_one_or_three_rank_two_matrices_containing_the_planted_motion():
    rng = np.random.default_rng(5)
    for _ in range(200):
        f1, f2, R, t, _ = _two_views(rng, 7)
        Es = oracle.sevenpt(f1, f2)
        assert len(Es) in (1, 3)
        for E in Es:
            assert max(abs(a @ E @ b) for a, b in zip(f1, f2)) < 1e-10
            assert abs(np.linalg.det(E / np.linalg.norm(E))) < 1e-10
        assert _nearest(Es, _true_essential(R, t)) < 1e-7


def test_ransac_finds_the_planted_inliers():
    rng = np.random.default_rng(3)
    f1, f2, R, t, good = _two_views(rng, 500, noise_deg=0.05, outlier_frac=0.3)
    thr = 2.0 * (1.0 - np.cos(np.deg2rad(1.0)))
    r = oracle.ransac_rel_pose(f1, f2, thr, 400, seed=5, want_counts=True)
    assert r["status"] == 0 and r["n_inliers"] >= 0.9 * good.sum()
    ang = np.arccos(np.clip((np.trace(r["T"][:, :3].T @ R) - 1) / 2, -1, 1))
    tdir = np.arccos(np.clip(r["T"][:, 3] @ t / np.linalg.norm(t), -1, 1))
    assert ang < np.deg2rad(1.0) and tdir < np.deg2rad(5.0)
    ra = oracle.ransac_rel_pose(f1, f2, thr, 5000, seed=5, adaptive=True)
    assert ra["iters_used"] < 5000 and ra["n_inliers"] >= 0.9 * good.sum()
    assert oracle.ransac_rel_pose(f1[:7], f2[:7], thr, 20)["status"] == 1      # fewer than 8 correspondences
    for algorithm, need in ((5, 8), (7, 9)):
        r = oracle.ransac_rel_pose(f1, f2, thr, 200, seed=6, algorithm=algorithm, want_counts=True)
        assert r["status"] == 0 and r["n_inliers"] >= 0.9 * good.sum()
        ang = np.arccos(np.clip((np.trace(r["T"][:, :3].T @ R) - 1) / 2, -1, 1))
        assert ang < np.deg2rad(1.0)
        assert (r["counts"] >= 0).mean() > 0.95                                 # nearly every sample yields a model
        assert oracle.ransac_rel_pose(f1[:need - 1], f2[:need - 1], thr, 20, algorithm=algorithm)["status"] == 1
