"""The generalised P3P restatement (oracle/gp3p_core.h) checked by means that do not share its code:
planted poses must be among the solutions; every returned pose must satisfy the three ray constraints (numpy
evaluation); the real positive roots of the octic found by numpy.roots (companion-matrix eigenvalues) must be the
depths of the returned solutions; the one-camera case must agree with Kneip's central P3P of the oracle; RANSAC with
GP3P hypotheses must find the planted inliers of a contaminated two-mirror problem."""
import numpy as np

import oracle
import synth


def _problem(rng, central=False, z_offsets=(150.0, -50.0)):
    """Three rays of a rig whose cameras sit on the z axis (the SOS rig: two mirror foci), a planted body pose."""
    R, t = synth.random_pose(rng, max_t=200.0, max_deg=30.0)
    cams = rng.integers(0, 2, 3) if not central else np.zeros(3, dtype=int)
    o = np.array([[0.0, 0.0, z_offsets[c]] for c in cams])
    while True:
        depth = rng.uniform(600.0, 6000.0, 3)
        d = rng.normal(size=(3, 3))
        fb = d / np.linalg.norm(d, axis=1, keepdims=True)
        X = o + depth[:, None] * fb                       # points in the body frame
        if np.linalg.norm(np.cross(X[1] - X[0], X[2] - X[0])) > 1e3:
            break
    P = X @ R.T + t                                       # P = R x + t
    return fb, o, P, R, t, depth


def test_planted_pose_is_among_the_solutions_and_all_solutions_are_consistent():
    rng = np.random.default_rng(21)
    found, total = 0, 0
    nsol_hist = np.zeros(9, dtype=int)
    for trial in range(300):
        fb, o, P, R, t, depth = _problem(rng, central=(trial % 5 == 0))
        sols, oct_ = oracle.gp3p(fb, o, P, want_octic=True)
        nsol_hist[len(sols)] += 1
        total += 1
        errs = [max(np.abs(T[:, :3] - R).max(), np.abs(T[:, 3] - t).max() / 1e3) for T in sols]
        found += int(len(errs) > 0 and min(errs) < 1e-7)
        for T in sols:                                    # every solution: x_i = R^T (P_i - t) lies on ray i, in front of it
            Rk, tk = T[:, :3], T[:, 3]
            assert np.allclose(Rk @ Rk.T, np.eye(3), atol=1e-9) and np.linalg.det(Rk) > 0.999
            x = (P - tk) @ Rk
            lam = np.einsum("ij,ij->i", x - o, fb)
            assert (lam > 0).all()
            assert np.abs(x - (o + lam[:, None] * fb)).max() < 1e-5 * np.abs(P).max()
        # independent root finder on the same octic: its real positive roots are first depths of solutions or were
        # rightly rejected (other depths non-positive); never fewer solutions than planted
        L = max(np.linalg.norm(P[i] - P[(i + 1) % 3]) for i in range(3))
        r = np.roots(oct_[::-1])
        real = r[np.abs(r.imag) < 1e-7 * (1 + np.abs(r.real))].real
        lam1 = sorted(float(np.dot(((P - T[:, 3]) @ T[:, :3])[0] - o[0], fb[0])) / L for T in sols)
        for v in lam1:
            assert np.min(np.abs(real - v)) < 1e-5 * max(1.0, abs(v)), (v, real)
        assert np.min(np.abs(real - depth[0] / L)) < 1e-6 * max(1.0, depth[0] / L)
    assert found >= total - 1, (found, total)             # (a planted solution may coincide with a double root)
    assert nsol_hist[2:].sum() > total // 5               # the problem does have several solutions, often


def test_one_camera_case_agrees_with_central_p3p():
    rng = np.random.default_rng(5)
    for _ in range(50):
        fb, o, P, R, t, depth = _problem(rng, central=True, z_offsets=(0.0, 0.0))
        sols = oracle.gp3p(fb, np.zeros((3, 3)), P)
        Rk, Ck = oracle.p3p_kneip(fb, P)
        n_checked = 0
        for Rw, Cw in zip(Rk, Ck):   # every central P3P pose that puts the three points ON their rays, in front of the
            x = (P - Cw) @ Rw        # camera, is a GP3P pose (Kneip's quartic also returns mirrored configurations)
            lam = np.einsum("ij,ij->i", x, fb)
            if (lam > 0).all() and np.abs(x - lam[:, None] * fb).max() < 1e-9 * np.abs(P).max():
                T = np.concatenate([Rw, Cw[:, None]], axis=1)
                assert min(np.abs(S - T).max() for S in sols) < 1e-6 * np.abs(P).max()
                n_checked += 1
        assert n_checked >= 1


def test_ransac_with_gp3p_finds_the_planted_inliers():
    rng = np.random.default_rng(8)
    pr = synth.make_abs_pose_problem(rng, 600, inlier_frac=0.5, noise_deg=0.1, noncentral=True)
    r = oracle.ransac_abs_pose(pr["f"], pr["p"], synth.THR_5DEG, 300, seed=3, cam=pr["cam"], cam_off=pr["cam_off"],
                               cam_rot=pr["cam_rot"], gp3p=True, want_counts=True)
    assert r["status"] == 0 and r["n_inliers"] >= 280
    assert synth.pose_error(r["T"], pr["R"], pr["t"])[0] < np.deg2rad(3.0)
    assert (r["counts"] >= 0).sum() > 100                  # (samples with outliers often have no real solution in front of the cameras)
    one = oracle.ransac_abs_pose(pr["f"], pr["p"], synth.THR_5DEG, 300, seed=3, cam=pr["cam"], cam_off=pr["cam_off"],
                                 cam_rot=pr["cam_rot"])
    assert not np.array_equal(one["T"], r["T"])            # a different hypothesis set than the one-camera P3P mode
