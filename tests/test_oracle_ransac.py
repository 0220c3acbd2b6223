"""CPU: pin the RANSAC / P3P / LM oracle with noise-free known-pose problems (the reference has
no golden vectors for OpenGV's output) and check the sampler's invariants."""
import numpy as np

import oracle
import synth


def test_quartic_real_roots():
    rng = np.random.default_rng(0)
    for _ in range(300):
        r = np.sort(rng.uniform(-2, 2, 4))
        if np.min(np.diff(r)) < 1e-2:
            continue
        got = np.sort(oracle.quartic_real_roots(np.poly(r) * rng.uniform(0.1, 10)))
        assert got.shape == (4,) and np.allclose(got, r, atol=1e-8)
    # two real + one complex pair
    a = np.convolve(np.poly([0.3, -1.2]), [1, -2 * 0.5, 0.25 + 0.7 ** 2])
    assert np.allclose(np.sort(oracle.quartic_real_roots(a)), [-1.2, 0.3], atol=1e-10)
    # no real root
    assert oracle.quartic_real_roots([1, 0, 2, 0, 5]).size == 0
    # biquadratic
    assert np.allclose(np.sort(oracle.quartic_real_roots([1, 0, -5, 0, 4])), [-2, -1, 1, 2], atol=1e-12)


def test_p3p_recovers_known_pose():
    rng = np.random.default_rng(1)
    errs = []
    for _ in range(300):
        pr = synth.make_abs_pose_problem(rng, 3, inlier_frac=1.0, noise_deg=0.0, noncentral=False)
        Rs, Cs = oracle.p3p_kneip(pr["f"], pr["p"])
        assert 1 <= len(Rs) <= 4
        for R in Rs:
            assert np.allclose(R @ R.T, np.eye(3), atol=1e-8) and np.linalg.det(R) > 0
        errs.append(min(max(np.abs(R - pr["R"]).max(), np.abs(C - pr["t"]).max() / np.linalg.norm(pr["t"]))
                        for R, C in zip(Rs, Cs)))
    errs = np.array(errs)
    assert np.median(errs) < 1e-11 and np.percentile(errs, 95) < 1e-8


def test_sampler_distinct_and_same_camera():
    rng = np.random.default_rng(2)
    pr = synth.make_abs_pose_problem(rng, 50, inlier_frac=1.0, noise_deg=0.0, noncentral=True, n_top=20)
    seen = set()
    errs = []
    for it in range(400):
        ok, T, s = oracle.hypothesis_once(pr["f"], pr["p"], 99, it, pr["cam"], pr["cam_off"], pr["cam_rot"])
        assert len(set(s.tolist())) == 4 and s.min() >= 0 and s.max() < 50
        assert len({int(pr["cam"][i]) for i in s[:3]}) == 1
        seen.add(tuple(s.tolist()))
        if ok:
            errs.append(max(synth.pose_error(T, pr["R"], pr["t"])))
    assert len(seen) > 390
    errs = np.array(errs)  # minimal samples can be ill-conditioned: bound the bulk, not the tail
    assert len(errs) > 390 and np.median(errs) < 1e-9 and np.percentile(errs, 95) < 1e-5


def test_ransac_noise_free_is_exact_and_deterministic():
    rng = np.random.default_rng(3)
    for nc in (False, True):
        pr = synth.make_abs_pose_problem(rng, 200, inlier_frac=1.0, noise_deg=0.0, noncentral=nc)
        kw = dict(cam=pr["cam"], cam_off=pr["cam_off"], cam_rot=pr["cam_rot"])
        r1 = oracle.ransac_abs_pose(pr["f"], pr["p"], synth.THR_5DEG, 20, seed=5, **kw)
        r2 = oracle.ransac_abs_pose(pr["f"], pr["p"], synth.THR_5DEG, 20, seed=5, **kw)
        assert r1["status"] == 0 and r1["n_inliers"] == 200 and r1["mask"].all()
        assert np.array_equal(r1["T"], r2["T"])
        ang, terr = synth.pose_error(r1["T"], pr["R"], pr["t"])
        assert ang < 1e-6 and terr < 1e-6  # reference bar: pose within 1e-6 rel


def test_ransac_with_outliers_and_adaptive_stop():
    rng = np.random.default_rng(4)
    pr = synth.make_abs_pose_problem(rng, 1500, inlier_frac=0.35, noise_deg=0.2, noncentral=True)
    kw = dict(cam=pr["cam"], cam_off=pr["cam_off"], cam_rot=pr["cam_rot"])
    full = oracle.ransac_abs_pose(pr["f"], pr["p"], synth.THR_5DEG, 600, seed=11, want_counts=True, **kw)
    assert full["iters_used"] == 600
    assert full["n_inliers"] >= 0.9 * pr["is_inlier"].sum()
    assert full["counts"].max() == full["n_inliers"] and full["counts"][full["best_iter"]] == full["n_inliers"]
    assert np.argmax(full["counts"]) == full["best_iter"]  # first maximum wins
    scores = oracle.score_points(pr["f"], pr["p"], full["T"], **kw)
    assert np.array_equal(scores < synth.THR_5DEG, full["mask"])
    ad = oracle.ransac_abs_pose(pr["f"], pr["p"], synth.THR_5DEG, 600, seed=11, adaptive=True, **kw)
    assert ad["iters_used"] < 600
    first = full["counts"][: ad["iters_used"]]
    assert ad["best_iter"] == int(np.argmax(first)) and ad["n_inliers"] == first.max()
    # too few points -> no model
    none = oracle.ransac_abs_pose(pr["f"][:3], pr["p"][:3], synth.THR_5DEG, 10, seed=1)
    assert none["status"] == 1 and none["n_inliers"] == 0 and np.array_equal(none["T"][:, :3], np.eye(3))


def test_refine_converges_to_known_pose():
    rng = np.random.default_rng(5)
    for nc in (False, True):
        pr = synth.make_abs_pose_problem(rng, 300, inlier_frac=1.0, noise_deg=0.0, noncentral=nc)
        kw = dict(cam=pr["cam"], cam_off=pr["cam_off"], cam_rot=pr["cam_rot"])
        T0 = np.hstack([pr["R"] @ synth.rot_from_axis_angle([1, 2, 3], 0.02), (pr["t"] + [5., -3, 4])[:, None]])
        T1, cost, its = oracle.refine_abs_pose(pr["f"], pr["p"], T0, max_lm_iter=200, **kw)
        a0, t0 = synth.pose_error(T0, pr["R"], pr["t"])
        a1, t1 = synth.pose_error(T1, pr["R"], pr["t"])
        assert a1 < 1e-2 * a0 and t1 < 1e-2 * t0 and cost < 1e-18
        idx = np.arange(0, 300, 2, dtype=np.int32)
        T2, _, _ = oracle.refine_abs_pose(pr["f"], pr["p"], T0, idx=idx, max_lm_iter=50, **kw)
        T3, _, _ = oracle.refine_abs_pose(pr["f"][idx], pr["p"][idx], T0, max_lm_iter=50,
                                          cam=None if pr["cam"] is None else pr["cam"][idx],
                                          cam_off=pr["cam_off"], cam_rot=pr["cam_rot"])
        assert np.array_equal(T2, T3)


def test_epnp_recovers_known_poses_and_sampler_is_distinct():
    """EPnP restatement (oracle/epnp_core.h): noise-free 5..8-point problems give the planted pose to 1e-9; the
    6-index sampler returns distinct in-range indices, deterministic in (seed, iteration); RANSAC with EPnP
    hypotheses finds the planted inliers of a contaminated central problem."""
    rng = np.random.default_rng(3)
    worst = 0.0
    for trial in range(200):
        n = int(rng.integers(5, 9))
        P = rng.uniform(-2, 2, (n, 3)) + np.array([0.0, 0.0, 5.0])
        R, t = synth.random_pose(rng)
        pc = (P - t) @ R
        if (pc[:, 2] <= 0.2).any():
            continue
        f = pc / np.linalg.norm(pc, axis=1, keepdims=True)
        T = oracle.epnp(f, P)
        assert T is not None
        worst = max(worst, np.abs(T[:, :3] - R).max(), np.abs(T[:, 3] - t).max())
    assert worst < 2e-7, worst   # (Jacobi sweeps stop at off / diag < 1e-13, ORC_JACOBI12_TOL: 6e-8 here; 1e-9 with 1e-20)
    assert oracle.epnp(np.ones((4, 3)), np.ones((4, 3))) is None and oracle.sample_distinct(5, 6, 1, 0) is None
    for it in range(50):
        s = oracle.sample_distinct(9, 6, 123, it)
        assert len(set(s.tolist())) == 6 and s.min() >= 0 and s.max() < 9
        assert np.array_equal(s, oracle.sample_distinct(9, 6, 123, it))
    seen = {tuple(oracle.sample_distinct(6, 6, 5, it)) for it in range(20)}
    assert all(sorted(s) == list(range(6)) for s in seen) and len(seen) > 5
    # 6-point samples: 0.7^6 = 12 % of the samples are outlier-free
    pr = synth.make_abs_pose_problem(rng, 500, inlier_frac=0.7, noise_deg=0.1, noncentral=False)
    r = oracle.ransac_abs_pose(pr["f"], pr["p"], synth.THR_5DEG, 200, seed=4, epnp=True)
    assert r["status"] == 0 and r["n_inliers"] >= 340 and synth.pose_error(r["T"], pr["R"], pr["t"])[0] < np.deg2rad(3.0)
    ra = oracle.ransac_abs_pose(pr["f"], pr["p"], synth.THR_5DEG, 2000, seed=4, epnp=True, adaptive=True)
    assert ra["iters_used"] < 2000 and ra["n_inliers"] >= 340


def test_eigen_solvers_against_numpy_eigh_and_each_other():
    """The 12 x 12 eigen-solver EPnP uses (round-robin Jacobi) and the independent Householder + implicit-QL solver of
    the oracle, both against numpy.linalg.eigh and against each other (invariant subspaces of the four smallest
    eigenvalues).  Random symmetric matrices, Gram matrices with a near-null space like EPnP's M^T M, repeated
    eigenvalues, a diagonal and the zero matrix."""
    rng = np.random.default_rng(11)
    cases = []
    for _ in range(40):
        B = rng.normal(size=(12, 12))
        cases.append(B + B.T)
    for _ in range(40):                                  # rank-deficient Gram matrices + noise
        M = rng.normal(size=(12, 12)) * rng.uniform(0.01, 100.0)
        M[:, 8:] = M[:, :4] @ rng.normal(size=(4, 4)) + 1e-7 * rng.normal(size=(12, 4))
        cases.append(M.T @ M)
    Q, _ = np.linalg.qr(rng.normal(size=(12, 12)))
    cases.append(Q @ np.diag([1, 1, 1, 2, 2, 3, 3, 3, 3, 5, 8, 8.0]) @ Q.T)
    cases.append(np.diag(np.arange(12.0)))
    cases.append(np.zeros((12, 12)))
    for A in cases:
        A = 0.5 * (A + A.T)
        got = oracle.symeig12(A)
        assert got is not None
        w = np.linalg.eigvalsh(A)
        scale = max(1.0, np.abs(w).max())
        for d, V in (got, oracle.jacobi12(A)):
            assert np.allclose(np.sort(d), w, rtol=0, atol=1e-12 * scale)
            assert np.allclose(V.T @ V, np.eye(12), atol=1e-12)                 # orthonormal columns
            assert np.allclose(A @ V, V * d[None, :], atol=1e-11 * scale)       # A v_k = d_k v_k


def test_twopt_translation_from_a_known_rotation():
    """TWOPT: pure translation (the rotation is the binding's identity prior) from 2-point samples; the planted translation
    is recovered, RANSAC counts the planted inliers, and a scipy least-squares fit of the same two-ray problem agrees."""
    from scipy.optimize import least_squares
    rng = np.random.default_rng(17)
    n = 400
    t_true = np.array([120.0, -40.0, 65.0])
    P = rng.normal(size=(n, 3))
    P = P / np.linalg.norm(P, axis=1, keepdims=True) * rng.uniform(800, 6000, (n, 1))
    f = P - t_true
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    f = synth.perturb_bearings(rng, f, 0.05)
    bad = rng.random(n) < 0.4
    g = rng.normal(size=(n, 3))
    f[bad] = (g / np.linalg.norm(g, axis=1, keepdims=True))[bad]
    r = oracle.ransac_abs_pose(f, P, synth.THR_5DEG, 100, seed=2, twopt=True, want_counts=True)
    assert r["status"] == 0 and np.array_equal(r["T"][:, :3], np.eye(3))
    assert np.linalg.norm(r["T"][:, 3] - t_true) < 60.0 and r["n_inliers"] >= 0.95 * (~bad).sum()
    # one clean 2-point sample against an independent solver of the same least-squares problem
    i, j = np.flatnonzero(~bad)[:2]
    res = least_squares(lambda t: np.concatenate([(np.eye(3) - np.outer(f[k], f[k])) @ (P[k] - t) for k in (i, j)]), np.zeros(3))
    one = oracle.ransac_abs_pose(f[[i, j]], P[[i, j]], 1.0, 1, seed=0, twopt=True)
    assert np.allclose(one["T"][:, 3], res.x, rtol=1e-9, atol=1e-6)
    ra = oracle.ransac_abs_pose(f, P, synth.THR_5DEG, 2000, seed=2, twopt=True, adaptive=True)
    assert ra["iters_used"] < 60                             # 1 - w^2 with w = 0.6: a handful of iterations


def test_refinement_reaches_the_minimum_scipy_least_squares_finds():
    """K10 (optimize_nonlinear) by an independent optimiser: scipy's Levenberg-Marquardt (MINPACK, numerical Jacobian, tolerances
    1e-15) on the same residual r_i = 1 - f_i . u_i / |u_i|, u_i = R^T (p_i - t) - o_i, in the same (t, Cayley) parametrisation,
    from the same start, on noisy correspondences: the oracle's analytic-Jacobian LM with its second-order term ends at a cost
    no higher than scipy's (measured: 1e-5 .. 1e-6 LOWER, in 5 iterations) and at the same pose (7e-5 degrees, 0.06 mm)."""
    from scipy.optimize import least_squares

    def cay2rot(c):
        x, y, z = c
        s = 1 + x * x + y * y + z * z
        return np.array([[1 + x * x - y * y - z * z, 2 * (x * y - z), 2 * (x * z + y)],
                         [2 * (x * y + z), 1 - x * x + y * y - z * z, 2 * (y * z - x)],
                         [2 * (x * z - y), 2 * (y * z + x), 1 - x * x - y * y + z * z]]) / s

    def rot2cay(R):
        A = (R - np.eye(3)) @ np.linalg.inv(R + np.eye(3))
        return np.array([A[2, 1], A[0, 2], A[1, 0]])

    rng = np.random.default_rng(11)
    for nc in (True, False):
        n = 600
        pr = synth.make_abs_pose_problem(rng, n, inlier_frac=1.0, noise_deg=0.3, noncentral=nc)
        kw = dict(cam=pr["cam"], cam_off=pr["cam_off"], cam_rot=pr["cam_rot"])
        T0 = np.hstack([pr["R"] @ synth.rot_from_axis_angle([1, 2, 3], 0.01), (pr["t"] + [3., -2, 4])[:, None]])
        T1, cost, its = oracle.refine_abs_pose(pr["f"], pr["p"], T0, max_lm_iter=30, **kw)
        o = pr["cam_off"][pr["cam"]] if nc else np.zeros((n, 3))

        def res(x):
            u = (pr["p"] - x[:3]) @ cay2rot(x[3:]) - o
            return 1.0 - (pr["f"] * u).sum(1) / np.linalg.norm(u, axis=1)

        sol = least_squares(res, np.concatenate([T0[:, 3], rot2cay(T0[:, :3])]), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15,
                            max_nfev=4000)
        c_scipy = float((res(sol.x) ** 2).sum())
        c_oracle = float((res(np.concatenate([T1[:, 3], rot2cay(T1[:, :3])])) ** 2).sum())
        assert abs(c_oracle - cost) <= 1e-9 * cost                   # (the cost the oracle reports is this cost)
        assert c_oracle <= c_scipy * (1.0 + 1e-4), (c_oracle, c_scipy)
        Rs = cay2rot(sol.x[3:])
        ang = np.degrees(np.arccos(np.clip((np.trace(Rs.T @ T1[:, :3]) - 1) / 2, -1, 1)))
        assert ang < 1e-3 and np.linalg.norm(sol.x[:3] - T1[:, 3]) < 0.5, (ang, np.linalg.norm(sol.x[:3] - T1[:, 3]))
        assert its <= 10
