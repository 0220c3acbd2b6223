"""The rounding bound behind ransac_score_kernel's single-precision first tier (csrc/ransac.hip: score_tier1_lane,
score_tier1_hyp, the `tier1` lambda), checked numerically on the CPU: a numpy model of the tier
(float32 arithmetic with fused multiply-adds, the constants as the kernel computes them) against the reference's
decision in double precision (the oracle's formula, operation for operation) on tens of millions of (hypothesis, point)
cases built to be hard -- bearings on the threshold cone, hypotheses that put the camera centre next to a point
(cancellation), matrices that are not rotations, units from 1e-12 to 1e8, bearings not of unit length.  A case the
tier DECIDES must agree with the reference; what it leaves undecided goes to double precision on the GPU.  The GPU
tests compare the kernel's counts with the oracle's; this one tests the argument itself, independent of the kernel."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32
EPS = 2.0 ** -24
K = 1048576.0
QBAND = 3.47 / K + 20.0 * EPS
E2 = (3.47 * K + 6.0) * 78.03 * EPS * EPS * (1.0 + 1e-4)
MAXMAG = 1e9


def test_constants_are_the_kernels():
    src = open(os.path.join(ROOT, "vo_single_camera_sos_amd", "csrc", "ransac.hip")).read()
    for text in ("kTier1K = 1048576.0", "kTier1Eps = 5.9604644775390625e-08", "kTier1QBand = 3.47 / kTier1K + 20.0 * kTier1Eps",
                 "kTier1E2 = (3.47 * kTier1K + 6.0) * 78.03 * kTier1Eps * kTier1Eps * (1.0 + 1e-4)", "kTier1Max = 1e9",
                 "(c2 + kTier1QBand) * (1.0 + 1e-6)", "(c2 - kTier1QBand) * (1.0 - 1e-6)", "(p1 * p1 + om * om) * (1.0 + 1e-6)",
                 "kTier1E2 * fmax(rho * rho, 1.0) * (1.0 + 1e-6)", "kTier1E2 * (im * im) * (1.0 + 1e-6) + 1e-24",
                 "fabs(fn - 1.0) <= 1e-6"):
        assert text in src, "the kernel no longer computes %r: update the model in this test with it" % text
    assert float(F32(EPS)) == 5.9604644775390625e-08


def fma32(a, b, c):
    """float32 fused multiply-add: the product of two floats is exact in double precision, the sum is rounded once there
    and once more to float (a double rounding that differs from the true fma in ~1e-9 of the cases by one ulp: inside
    what the bound allows for, and this is a model)."""
    with np.errstate(all="ignore"):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def tier1(f, p, o, h, thr):
    """f, p, o [n,3] f64; h [m,12] f64 (R column-major, i) -> hi, lo [m,n] bool (decided inlier / decided not)."""
    c2 = (1.0 - thr) ** 2
    c2h, c2l = F32((c2 + QBAND) * (1.0 + 1e-6)), F32((c2 - QBAND) * (1.0 - 1e-6))
    with np.errstate(all="ignore"):
        p1 = np.abs(p).sum(1)
        om = np.abs(o).max(1)
        fn = (f * f).sum(1)
        L = np.where((np.abs(fn - 1.0) <= 1e-6) & (p1 <= MAXMAG) & (om <= MAXMAG), ((p1 * p1 + om * om) * (1.0 + 1e-6)), np.inf).astype(F32)
        rho = np.abs(h[:, :9]).max(1)
        im = np.abs(h[:, 9:]).max(1)
        fin = np.isfinite(h).all(1) & (rho <= MAXMAG) & (im <= MAXMAG)
        ea = np.where(fin, E2 * np.maximum(rho * rho, 1.0) * (1.0 + 1e-6), np.nan).astype(F32)[:, None]
        eb = np.where(fin, E2 * im * im * (1.0 + 1e-6) + 1e-24, np.nan).astype(F32)[:, None]
        P, Fv, O, H = p.astype(F32), f.astype(F32), o.astype(F32), h.astype(F32)
        u = []
        for k in range(3):
            c = (H[:, 9 + k, None] - O[None, :, k]).astype(F32)
            t = fma32(H[:, 6 + k, None], P[None, :, 2], c)
            t = fma32(H[:, 3 + k, None], P[None, :, 1], t)
            u.append(fma32(H[:, k, None], P[None, :, 0], t))
        s = fma32(Fv[None, :, 0], u[0], fma32(Fv[None, :, 1], u[1], (Fv[None, :, 2] * u[2]).astype(F32)))
        q = fma32(u[0], u[0], fma32(u[1], u[1], (u[2] * u[2]).astype(F32)))
        lhs = (s * np.abs(s)).astype(F32)    # the signed square
        e2 = fma32(ea, L[None, :], eb)
        hi = lhs > fma32(np.broadcast_to(c2h, q.shape), q, e2)
        lo = lhs < fma32(np.broadcast_to(c2l, q.shape), q, -e2)
    return hi, lo


def exact(f, p, o, h, thr):
    """The reference's decision (oracle/ransac_core.h sv_score: no contraction) -> [m,n] bool."""
    with np.errstate(all="ignore"):
        w = []
        for k in range(3):
            v = ((h[:, k, None] * p[None, :, 0] + h[:, 3 + k, None] * p[None, :, 1]) + h[:, 6 + k, None] * p[None, :, 2]) + h[:, 9 + k, None]
            w.append(v - o[None, :, k])
        nrm = np.sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2])
        g = [wk / nrm for wk in w]
        return (1.0 - ((f[None, :, 0] * g[0] + f[None, :, 1] * g[1]) + f[None, :, 2] * g[2])) < thr


def rot(rng, ang):
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def make_case(rng, n, scale, thr_deg, f_scale=1.0):
    """World points, two camera offsets, a true pose; half of the bearings ON the threshold cone of the true pose
    (+- 1e-10 .. 1e-2 rad), the rest inliers with noise and outliers."""
    p = rng.normal(size=(n, 3))
    p *= (rng.uniform(0.5, 6.0, (n, 1)) * 1000.0 / np.linalg.norm(p, axis=1, keepdims=True))
    o = np.where(rng.random((n, 1)) < 0.5, np.array([[0.0, 0.0, 60.0]]), np.array([[0.0, 0.0, -70.0]]))
    Rt, it = rot(rng, np.deg2rad(rng.uniform(0, 6))), rng.normal(size=3) * 40.0
    u = p @ Rt.T + it - o
    f = u / np.linalg.norm(u, axis=1, keepdims=True)
    k = n // 2
    ang = np.deg2rad(thr_deg) + rng.choice([-1.0, 1.0], k) * 10.0 ** rng.uniform(-10, -2, k)
    ang[: k // 10] = np.deg2rad(thr_deg)
    ax = np.cross(f[:k], rng.normal(size=(k, 3)))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    f[:k] = f[:k] * np.cos(ang)[:, None] + np.cross(ax, f[:k]) * np.sin(ang)[:, None]
    f[k:] += rng.normal(size=(n - k, 3)) * np.deg2rad(0.3)
    out = rng.random(n) < 0.1
    f[out] = rng.normal(size=(int(out.sum()), 3))
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    return f * f_scale, p * scale, o * scale, Rt, it * scale


def hypotheses(rng, Rt, it, p, o, m):
    """R (column-major) and i for m hypotheses: the true pose (the cone!), small and large perturbations of it, camera
    centres placed next to world points (u = R p + i - o nearly cancels), matrices that are not rotations."""
    hs = []
    ext = float(np.abs(p).max())
    for j in range(m):
        kind = j % 8
        R, i = Rt, it
        if kind == 1:
            R, i = rot(rng, 10.0 ** rng.uniform(-9, -2)) @ Rt, it + rng.normal(size=3) * ext * 10.0 ** rng.uniform(-9, -3)
        elif kind == 2:
            R, i = rot(rng, rng.uniform(0, 0.3)) @ Rt, it + rng.normal(size=3) * ext * 0.02
        elif kind == 3:
            R, i = rot(rng, rng.uniform(0, np.pi)), rng.normal(size=3) * ext
        elif kind == 4:      # the camera centre next to a world point: u cancels to 1e-1 .. 1e-9 of |p|
            q = rng.integers(len(p))
            R = rot(rng, rng.uniform(0, np.pi))
            i = -(R @ p[q]) + o[q] + rng.normal(size=3) * ext * 10.0 ** rng.uniform(-9, -1)
        elif kind == 5:      # not a rotation
            R = (rot(rng, rng.uniform(0, np.pi)) + rng.normal(size=(3, 3)) * 10.0 ** rng.uniform(-6, 0)) * 10.0 ** rng.uniform(-2, 2)
            i = rng.normal(size=3) * ext
        elif kind == 6:      # far away / huge
            R, i = rot(rng, rng.uniform(0, np.pi)), rng.normal(size=3) * ext * 10.0 ** rng.uniform(1, 12)
        elif kind == 7:
            R, i = Rt * (1.0 + 10.0 ** rng.uniform(-9, -4)), it
        hs.append(np.concatenate([R.T.reshape(9), i]))   # column-major: h[k + 3 l] = R[k, l]
    return np.array(hs)


def test_decided_cases_agree_with_the_reference():
    rng = np.random.default_rng(2026)
    total = decided = 0
    typical_und = []
    for scale, thr_deg, f_scale in [(1.0, 5.0, 1.0), (1e-3, 5.0, 1.0), (1e-6, 5.0, 1.0), (1e3, 5.0, 1.0), (1e5, 5.0, 1.0),
                                    (1e-12, 5.0, 1.0), (1.0, 0.5, 1.0), (1.0, 20.0, 1.0), (1.0, 50.0, 1.0), (1.0, 5.0, 1.0 + 4e-7),
                                    (1.0, 5.0, 1.0 - 4e-7), (1.0, 5.0, 1.001), (1e-3, 1.0, 1.0)]:
        thr = 1.0 - np.cos(np.deg2rad(thr_deg))
        assert thr < 0.5
        for rep in range(3):
            f, p, o, Rt, it = make_case(rng, 2048, scale, thr_deg, f_scale)
            h = hypotheses(rng, Rt, it, p, o, 256)
            hi, lo = tier1(f, p, o, h, thr)
            ex = exact(f, p, o, h, thr)
            assert not (hi & lo).any()
            bad_in, bad_out = hi & ~ex, lo & ex
            assert not bad_in.any() and not bad_out.any(), \
                "tier 1 decided against the reference: scale %g, threshold %g deg, %d + %d cases" % (
                    scale, thr_deg, bad_in.sum(), bad_out.sum())
            total += hi.size
            decided += int(hi.sum() + lo.sum())
            if f_scale == 1.0 and scale in (1.0, 1e-3) and thr_deg == 5.0:
                ok = np.arange(len(h)) % 8 == 2          # ordinary RANSAC hypotheses (a few degrees off)
                half = len(p) // 2                       # the points that do not sit on the true pose's cone
                typical_und.append(1.0 - (hi[ok][:, half:] | lo[ok][:, half:]).mean())
    assert total > 2e7
    # the tier is worth having: on ordinary hypotheses and points it leaves well under a thousandth undecided
    assert max(typical_und) < 1e-3, typical_und
    # and it is not vacuous on the hard set either (most of it is decidable; the cases with bearings of length 1.001 and the cone under near-true hypotheses are not)
    assert decided > 0.6 * total


def test_out_of_range_inputs_stay_undecided():
    rng = np.random.default_rng(7)
    f, p, o, Rt, it = make_case(rng, 256, 1.0, 5.0)
    h = hypotheses(rng, Rt, it, p, o, 64)
    thr = 1.0 - np.cos(np.deg2rad(5.0))
    for mutate in ("p_nan", "p_inf", "p_big", "f_len", "h_nan", "h_big", "tiny"):
        f2, p2, o2, h2 = f.copy(), p.copy(), o.copy(), h.copy()
        if mutate == "p_nan":
            p2[:, 1] = np.nan
        elif mutate == "p_inf":
            p2[:, 0] = np.inf
        elif mutate == "p_big":
            p2 *= 1e9
        elif mutate == "f_len":
            f2 *= 1.01
        elif mutate == "h_nan":
            h2[:, 4] = np.nan
        elif mutate == "h_big":
            h2[:, 9:] = 1e10
        elif mutate == "tiny":      # q underflows in float
            p2 *= 1e-25
            o2 *= 1e-25
            h2[:, 9:] = rng.normal(size=(len(h2), 3)) * 1e-23
        hi, lo = tier1(f2, p2, o2, h2, thr)
        assert not hi.any() and not lo.any(), mutate
