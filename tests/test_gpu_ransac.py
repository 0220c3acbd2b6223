"""GPU parity: sosvo_ransac_abs_pose / sosvo_refine_abs_pose (through the C ABI) against the
CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): inlier masks, inlier index lists, per-hypothesis inlier counts
and the winning iteration are integer work -> bit-exact.  The RANSAC pose is produced by the same
IEEE operations in the same order on both sides -> compared bit-exact too.  The LM refinement
defines its sums as 256-way interleaved partials combined by a tree on both sides, so the refined
pose is compared bit-exact as well (north_star only asks for rel-tol 1e-6)."""
import numpy as np
import pytest
import torch

import oracle
import synth

pytestmark = pytest.mark.gpu


def _pack(problems, S, noncentral):
    P = len(problems)
    f = np.zeros((P, S, 3))
    p = np.zeros((P, S, 3))
    cam = np.zeros((P, S), dtype=np.int32)
    n = np.zeros(P, dtype=np.int32)
    for b, pr in enumerate(problems):
        k = pr["f"].shape[0]
        n[b] = k
        f[b, :k] = pr["f"]
        p[b, :k] = pr["p"]
        if noncentral:
            cam[b, :k] = pr["cam"]
    return f, p, cam, n


def _to(dev, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


def _run_batch(ctx, problems, S, noncentral, max_iter, seed, adaptive=False, ident=True, thr=synth.THR_5DEG, epnp=False,
               gp3p=False, twopt=False):
    f, p, cam, n = _pack(problems, S, noncentral)
    dev = ctx.device
    tf, tp, tcam, tn = _to(dev, f, p, cam, n)
    kw = {}
    if noncentral:
        off, rot = _to(dev, problems[0]["cam_off"], problems[0]["cam_rot"])
        kw = dict(cam=tcam, cam_off=off, cam_rot=rot, cam_rot_identity=ident)
    out = ctx.ransac_abs_pose(tf, tp, tn, thr, max_iter, seed=seed, adaptive=adaptive, want_counts=True, epnp=epnp, gp3p=gp3p,
                              twopt=twopt, **kw)
    ctx.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    for b, pr in enumerate(problems):
        okw = dict(cam=pr["cam"], cam_off=pr["cam_off"], cam_rot=pr["cam_rot"]) if noncentral else {}
        want = oracle.ransac_abs_pose(pr["f"], pr["p"], thr, max_iter, seed=seed + b, adaptive=adaptive,
                                      want_counts=True, epnp=epnp, gp3p=gp3p, twopt=twopt, **okw)
        k = n[b]
        used = want["iters_used"]
        assert got["info"][b, 1] == used, "iterations drawn, problem %d" % b
        assert np.array_equal(got["counts"][b, :used], want["counts"][:used]), "hypothesis counts, problem %d" % b
        assert got["info"][b, 0] == want["best_iter"], "winning iteration, problem %d" % b
        assert got["info"][b, 2] == want["status"]
        assert np.array_equal(got["mask"][b, :k].astype(bool), want["mask"]), "inlier mask, problem %d" % b
        assert got["n_inliers"][b] == want["n_inliers"]
        assert np.array_equal(got["idx"][b, : want["n_inliers"]], np.nonzero(want["mask"])[0])
        assert np.array_equal(got["T"][b], want["T"]), "RANSAC pose bits, problem %d" % b
    return (tf, tp, tcam, tn, kw), out, got


def test_sqrt_div_are_ieee(ctx):
    # the bit-exactness argument rests on correctly rounded FP64 sqrt and divide on the device:
    # torch's sqrt/div kernels use the same device instructions sequences; check against numpy.
    rng = np.random.default_rng(0)
    a = np.abs(rng.normal(size=200000)) * 10.0 ** rng.integers(-30, 30, 200000)
    b = rng.normal(size=200000) * 10.0 ** rng.integers(-30, 30, 200000)
    ta, tb = _to(ctx.device, a, b)
    assert np.array_equal(torch.sqrt(ta).cpu().numpy(), np.sqrt(a))
    assert np.array_equal((ta / tb).cpu().numpy(), a / b)


def test_central_small_batch(ctx):
    rng = np.random.default_rng(1)
    probs = [synth.make_abs_pose_problem(rng, n, inlier_frac=fr, noise_deg=nz, noncentral=False)
             for n, fr, nz in [(64, 1.0, 0.0), (100, 0.5, 0.1), (257, 0.35, 0.2), (512, 0.8, 0.5), (5, 1.0, 0.0)]]
    _run_batch(ctx, probs, 512, False, 100, seed=1234)


def test_noncentral_ragged_batch_identity_and_general_path(ctx):
    rng = np.random.default_rng(2)
    probs = [synth.make_abs_pose_problem(rng, n, inlier_frac=0.4, noise_deg=0.2, noncentral=True, n_top=nt)
             for n, nt in [(300, 150), (1000, 100), (33, 30), (700, 699), (64, 0), (1024, 512)]]
    _run_batch(ctx, probs, 1024, True, 200, seed=77, ident=True)
    _run_batch(ctx, probs, 1024, True, 200, seed=77, ident=False)


def test_rotated_cameras_general_path(ctx):
    rng = np.random.default_rng(3)
    # rig whose cameras are rotated wrt the body: bearings are expressed in the rotated frames
    Rc = np.stack([synth.rot_from_axis_angle([0, 0, 1], 0.3), synth.rot_from_axis_angle([1, 1, 0], -0.2)])
    probs = []
    for n in (400, 800):
        pr = synth.make_abs_pose_problem(rng, n, inlier_frac=0.5, noise_deg=0.1, noncentral=True)
        pr["f"] = np.einsum("nji,nj->ni", Rc[pr["cam"]], pr["f"])  # f_cam = Rc^T f_body
        pr["cam_rot"] = Rc
        probs.append(pr)
    _, _, got = _run_batch(ctx, probs, 800, True, 150, seed=5, ident=False)
    for b, pr in enumerate(probs):
        assert got["n_inliers"][b] >= 0.9 * pr["is_inlier"].sum()


def test_degenerate_inputs(ctx):
    rng = np.random.default_rng(4)
    few = synth.make_abs_pose_problem(rng, 3, inlier_frac=1.0, noise_deg=0.0, noncentral=False)
    empty = dict(f=np.zeros((0, 3)), p=np.zeros((0, 3)), cam=None, cam_off=None, cam_rot=None)
    coll = synth.make_abs_pose_problem(rng, 50, inlier_frac=1.0, noise_deg=0.0, noncentral=False)
    coll["p"][:] = np.outer(np.linspace(1, 2, 50), [100.0, 200.0, 300.0])  # collinear world points
    _, _, got = _run_batch(ctx, [few, empty, coll], 64, False, 20, seed=9)
    assert got["info"][0, 2] == 1 and got["info"][1, 2] == 1 and got["info"][2, 2] == 1
    assert np.array_equal(got["T"][1][:, :3], np.eye(3))


def test_adaptive_stop_matches_sequential_semantics(ctx):
    rng = np.random.default_rng(5)
    probs = [synth.make_abs_pose_problem(rng, 900, inlier_frac=fr, noise_deg=0.2, noncentral=True)
             for fr in (0.3, 0.6, 0.95)]
    _, _, got = _run_batch(ctx, probs, 900, True, 500, seed=21, adaptive=True)
    assert got["info"][2, 1] < got["info"][0, 1] <= 500  # more inliers -> earlier stop


def test_c2_full_size(ctx):
    # BASELINE config 2: N = 4000 stacked top+bottom correspondences, 2000 iterations, 35 % inliers
    rng = np.random.default_rng(6)
    probs = [synth.make_abs_pose_problem(rng, 4000, inlier_frac=0.35, noise_deg=0.2, noncentral=True)
             for _ in range(3)]
    (tf, tp, tcam, tn, kw), out, got = _run_batch(ctx, probs, 4000, True, 2000, seed=2024)
    for b, pr in enumerate(probs):
        assert got["n_inliers"][b] >= 0.9 * pr["is_inlier"].sum()
    # K9 on the RANSAC inliers, compared with the oracle's LM started from the same pose
    T = out["T"].clone()
    kw2 = {k: v for k, v in kw.items() if k != "cam_rot_identity"}
    T, cost, iters = ctx.refine_abs_pose(tf, tp, tn, T, idx=out["idx"], m=out["n_inliers"], **kw2)
    ctx.synchronize()
    T = T.cpu().numpy()
    for b, pr in enumerate(probs):
        idx = got["idx"][b, : got["n_inliers"][b]]
        want, wcost, wit = oracle.refine_abs_pose(pr["f"], pr["p"], got["T"][b], idx=idx, cam=pr["cam"],
                                                  cam_off=pr["cam_off"], cam_rot=pr["cam_rot"])
        # north_star bar is rel-tol 1e-6; the summation order is part of the spec on both sides
        # (256-way interleaved partials + tree), so the LM trajectory is reproduced bit for bit
        assert np.array_equal(T[b], want), "refined pose, problem %d" % b
        assert wit == int(iters[b]) and wcost == float(cost[b])
        start = (oracle.score_points(pr["f"][idx], pr["p"][idx], got["T"][b], cam=pr["cam"][idx],
                                     cam_off=pr["cam_off"], cam_rot=pr["cam_rot"]) ** 2).sum()
        assert wcost <= start


def test_refine_noise_free_reaches_ground_truth(ctx):
    rng = np.random.default_rng(7)
    probs = [synth.make_abs_pose_problem(rng, 500, inlier_frac=1.0, noise_deg=0.0, noncentral=nc)
             for nc in (True, True)]
    f, p, cam, n = _pack(probs, 512, True)
    T0 = np.stack([np.hstack([pr["R"] @ synth.rot_from_axis_angle([1, 2, 3], 0.02),
                              (pr["t"] + [5., -3, 4])[:, None]]) for pr in probs])
    dev = ctx.device
    tf, tp, tcam, tn, tT, off, rot = _to(dev, f, p, cam, n, T0, probs[0]["cam_off"], probs[0]["cam_rot"])
    T, cost, iters = ctx.refine_abs_pose(tf, tp, tn, tT, cam=tcam, cam_off=off, cam_rot=rot, max_lm_iter=200)
    ctx.synchronize()
    T = T.cpu().numpy()
    for b, pr in enumerate(probs):
        want, _, _ = oracle.refine_abs_pose(pr["f"], pr["p"], T0[b], cam=pr["cam"], cam_off=pr["cam_off"],
                                            cam_rot=pr["cam_rot"], max_lm_iter=200)
        a1, t1 = synth.pose_error(T[b], pr["R"], pr["t"])
        a0, t0 = synth.pose_error(T0[b], pr["R"], pr["t"])
        assert a1 < 1e-2 * a0 and t1 < 1e-2 * t0
        assert np.array_equal(T[b], want)


def test_central_epnp_hypotheses_bit_exact(ctx):
    """SOSVO_FLAG_EPNP: 6-point samples solved by EPnP (Jacobi eigen-decompositions, Gauss-Newton on the betas, ...)
    reproduce the oracle's hypotheses bit for bit: per-hypothesis inlier counts, winning iteration, masks and pose;
    with and without the adaptive stop (w^6); too few points -> no model; the planted pose is recovered."""
    rng = np.random.default_rng(41)
    probs = [synth.make_abs_pose_problem(rng, n, inlier_frac=fr, noise_deg=nz, noncentral=False)
             for n, fr, nz in ((300, 0.6, 0.1), (700, 0.5, 0.2), (64, 0.9, 0.0), (5, 1.0, 0.0), (6, 1.0, 0.0))]
    for adaptive in (False, True):
        _, _, got = _run_batch(ctx, probs, 704, False, 300, 9, adaptive=adaptive, epnp=True)
        assert got["info"][3, 2] == 1 and got["n_inliers"][3] == 0          # 5 points: no 6-point sample
        assert got["info"][4, 2] == 0 and got["n_inliers"][4] == 6
        for b in (0, 1, 2):
            ang, terr = synth.pose_error(got["T"][b], probs[b]["R"], probs[b]["t"])
            assert ang < np.deg2rad(3.0), (b, ang)
    with pytest.raises(Exception):                                         # EPnP is for central problems
        pr = synth.make_abs_pose_problem(rng, 50, inlier_frac=1.0, noise_deg=0.0, noncentral=True)
        _run_batch(ctx, [pr], 64, True, 10, 1, epnp=True)


def test_gp3p_hypotheses_bit_exact(ctx):
    """SOSVO_FLAG_GP3P: samples of four correspondences across BOTH mirrors, generalised P3P (up to 8 poses), the fourth
    point picks one -- per-hypothesis inlier counts, winner, masks and pose bits equal the oracle's; fixed and adaptive
    iteration budgets; identity and rotated camera frames; the central case through the same solver."""
    rng = np.random.default_rng(31)
    probs = [synth.make_abs_pose_problem(rng, n, inlier_frac=fr, noise_deg=0.2, noncentral=True, n_top=nt)
             for n, nt, fr in [(300, 150, 0.5), (1000, 100, 0.35), (33, 30, 0.9), (700, 699, 0.6), (64, 0, 0.7), (1024, 512, 0.4)]]
    _, _, got = _run_batch(ctx, probs, 1024, True, 300, seed=11, ident=True, gp3p=True)
    for b, pr in enumerate(probs):
        assert got["info"][b, 2] == 0 and got["n_inliers"][b] >= 0.85 * pr["is_inlier"].sum(), b
    _run_batch(ctx, probs, 1024, True, 300, seed=11, ident=False, gp3p=True)
    _run_batch(ctx, probs, 1024, True, 2000, seed=12, adaptive=True, gp3p=True)
    Rc = np.stack([synth.rot_from_axis_angle([0, 0, 1], 0.3), synth.rot_from_axis_angle([1, 1, 0], -0.2)])
    rot = []
    for n in (400, 800):
        pr = synth.make_abs_pose_problem(rng, n, inlier_frac=0.5, noise_deg=0.1, noncentral=True)
        pr["f"] = np.einsum("nji,nj->ni", Rc[pr["cam"]], pr["f"])
        pr["cam_rot"] = Rc
        rot.append(pr)
    _run_batch(ctx, rot, 800, True, 150, seed=5, ident=False, gp3p=True)
    cen = [synth.make_abs_pose_problem(rng, n, inlier_frac=0.5, noise_deg=0.1, noncentral=False) for n in (200, 77)]
    _run_batch(ctx, cen, 256, False, 150, seed=6, gp3p=True)


def test_twopt_hypotheses_bit_exact(ctx):
    """SOSVO_FLAG_TWOPT: 2-point samples, translation from a known (identity) rotation; fixed and adaptive budgets."""
    rng = np.random.default_rng(41)
    probs = []
    for n in (300, 64, 1000):
        pr = synth.make_abs_pose_problem(rng, n, inlier_frac=0.6, noise_deg=0.1, noncentral=False)
        v = pr["p"] - pr["t"]                              # bearings of a camera that only translated
        f = synth.perturb_bearings(rng, v / np.linalg.norm(v, axis=1, keepdims=True), 0.1)
        pr["f"] = np.ascontiguousarray(np.where(pr["is_inlier"][:, None], f, pr["f"]))
        probs.append(pr)
    _, _, got = _run_batch(ctx, probs, 1024, False, 200, seed=3, twopt=True)
    for b, pr in enumerate(probs):
        assert np.array_equal(got["T"][b][:, :3], np.eye(3)) and got["n_inliers"][b] >= 0.9 * pr["is_inlier"].sum()
    _run_batch(ctx, probs, 1024, False, 2000, seed=4, adaptive=True, twopt=True)


def _threshold_cone_problem(rng, n, scale=1.0, f_scale=1.0, noncentral=True):
    """Inliers whose bearing error sits ON the 5-degree cone of the true pose, give or take 1e-9 .. 1e-3 rad (the band
    in which single precision cannot decide), plus ordinary inliers and outliers; world points scaled by `scale`
    (the tracker's threshold is an angle: units do not matter to the reference), bearings by `f_scale`."""
    pr = synth.make_abs_pose_problem(rng, n, inlier_frac=0.8, noise_deg=0.05, noncentral=noncentral)
    f = pr["f"]
    k = n // 2
    ang = np.deg2rad(5.0) + rng.choice([-1.0, 1.0], k) * 10.0 ** rng.uniform(-9, -3, k)
    ang[: k // 8] = np.deg2rad(5.0)
    ax = np.cross(f[:k], rng.normal(size=(k, 3)))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    f[:k] = f[:k] * np.cos(ang)[:, None] + np.cross(ax, f[:k]) * np.sin(ang)[:, None]
    order = rng.permutation(n)
    for key in ("f", "p") + (("cam",) if noncentral else ()):
        pr[key] = np.ascontiguousarray(pr[key][order])
    if noncentral:  # the hypothesis kernel samples per camera: keep the cameras in blocks as the front end delivers them
        o2 = np.argsort(pr["cam"], kind="stable")
        for key in ("f", "p", "cam"):
            pr[key] = np.ascontiguousarray(pr[key][o2])
        pr["cam_off"] = pr["cam_off"] * scale
    pr["p"] = pr["p"] * scale
    pr["f"] = pr["f"] * f_scale
    return pr


def test_score_single_precision_tier_on_the_threshold_cone_and_odd_inputs(ctx):
    """ransac_score_kernel's single-precision first tier (csrc/ransac.hip, score_tier1_*) may only decide where its
    rounding bound allows: counts bit-exact against the oracle where the decision is hardest (bearings on the cone),
    in other units (metres, kilometres, magnitudes beyond the tier's range and far below it), with bearings that are
    not of unit length, and with NaN / inf coordinates -- and identical to the double-precision-only form."""
    rng = np.random.default_rng(41)
    probs = [_threshold_cone_problem(rng, 700), _threshold_cone_problem(rng, 400, f_scale=1.0 + 3e-7),
             _threshold_cone_problem(rng, 300, f_scale=1.001), _threshold_cone_problem(rng, 300, f_scale=2.0)]
    bad = _threshold_cone_problem(rng, 500)
    bad["p"][7] = np.nan
    bad["p"][300, 1] = np.inf
    bad["f"][9, 2] = np.nan
    bad["p"][11] = 0.0
    bad["p"][12] = 1e200
    probs.append(bad)
    args, out, got = _run_batch(ctx, probs, 768, True, 300, seed=3, ident=True)
    for scale in (1e-3, 1e-6, 1e12, 1e-17):  # (a batch shares one rig: the camera offsets scale with the points)
        _run_batch(ctx, [_threshold_cone_problem(rng, 333, scale=scale), _threshold_cone_problem(rng, 130, scale=scale)], 384,
                   True, 300, seed=5, ident=True)
    central = [_threshold_cone_problem(rng, 640, noncentral=False), _threshold_cone_problem(rng, 90, scale=1e-3, noncentral=False)]
    _run_batch(ctx, central, 640, False, 300, seed=4)
    # the same launch without the tier: every output identical
    tf, tp, tcam, tn, kw = args
    ctx.set_hint_score_fp64_only(True)
    try:
        ref = ctx.ransac_abs_pose(tf, tp, tn, synth.THR_5DEG, 300, seed=3, want_counts=True, **kw)
        ctx.synchronize()
    finally:
        ctx.set_hint_score_fp64_only(False)
    for k in ("counts", "mask", "idx", "n_inliers", "info"):
        assert torch.equal(ref[k], out[k]), k
    assert np.array_equal(ref["T"].cpu().numpy(), got["T"], equal_nan=True)
    # (the cone really is populated: the exact pose leaves a good share of every problem within 1e-3 rad of it)
    assert got["n_inliers"][0] > 200
