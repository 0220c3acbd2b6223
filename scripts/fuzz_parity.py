#!/usr/bin/env python3
"""Differential fuzzing of the stage entry points against the CPU oracle (test infrastructure: the oracle is the checker).

    python scripts/fuzz_parity.py --minutes 10 --seed 1 [--stages median,gft,fast,agast,match,radius,orb,unwrap,ransac,describe,relpose,l2sort,pipeline,rgbd]

Every case draws its own sizes and parameters (image sizes around the strip / chunk / tile borders of the kernels, ragged
problem counts, empty masks, fractional minimum distances, budgets that end a selection inside a round, duplicate
descriptors, ...) from a generator seeded by (--seed, case number), runs the HIP stage through the C ABI and compares the
result with the oracle's bit for bit.  The first mismatch stops the run and prints the stage, the case number and the drawn
parameters: `--stages <stage> --first-case <n> --cases 1` with the same seed replays it.  Prints a progress line every ~20 s.
A case that only trips a documented capacity (status flags) counts as `flagged`, not as a mismatch.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from vo_single_camera_sos_amd import orb_pattern  # noqa: E402
from vo_single_camera_sos_amd.device import Context, SosvoError  # noqa: E402


def _dev(ctx, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device) for a in arrs]


def _image(rng, rows, cols, kind):
    """gray test images: noise, smooth texture, blobs on black, constant"""
    if kind == 0:
        return rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    if kind == 1:
        a = rng.normal(0, 1, (rows // 4 + 2, cols // 4 + 2))
        a = np.kron(a, np.ones((4, 4)))[:rows, :cols]
        return np.clip(128 + 60 * a + rng.normal(0, 6, (rows, cols)), 0, 255).astype(np.uint8)
    if kind == 2:
        img = np.zeros((rows, cols), np.uint8)
        for _ in range(int(rng.integers(1, 60))):
            y, x, s = int(rng.integers(0, rows)), int(rng.integers(0, cols)), int(rng.integers(1, 7))
            img[y:y + s, x:x + s] = rng.integers(30, 256)
        return img
    return np.full((rows, cols), int(rng.integers(0, 256)), np.uint8)


def _masks(rng, rows, cols, nmask):
    bits = np.zeros((rows, cols), np.uint32)
    kind = int(rng.integers(0, 4))
    if kind == 0:     # azimuthal sectors sharing their border columns, a few rows out
        edges = np.linspace(0, cols - 1, nmask + 1).astype(int)
        for m in range(nmask):
            bits[:, edges[m]:edges[m + 1] + 1] |= np.uint32(1 << m)
        bits[:int(rng.integers(0, 4))] = 0
    elif kind == 1:   # random rectangles (overlapping, possibly empty)
        for m in range(nmask):
            y0, y1 = sorted(rng.integers(0, rows, 2))
            x0, x1 = sorted(rng.integers(0, cols, 2))
            bits[y0:y1, x0:x1] |= np.uint32(1 << m)
    elif kind == 2:   # everything in every mask
        bits[:] = np.uint32((1 << nmask) - 1)
    else:             # random pixels
        for m in range(nmask):
            bits |= (rng.random((rows, cols)) < rng.random()).astype(np.uint32) << np.uint32(m)
    return bits


def fuzz_median(ctx, rng):
    k = int(rng.choice([3, 5, 11]))
    rows, cols, n = int(rng.integers(1, 90)), int(rng.integers(1, 260)), int(rng.integers(1, 4))
    if rng.random() < 0.08:
        rows, cols = int(rng.integers(90, 300)), int(rng.integers(260, 1500))
    img = rng.integers(0, 256, (n, rows, cols, 3), dtype=np.uint8)
    if rng.random() < 0.3:
        img[:, :, :, :] = (img >> int(rng.integers(4, 8))) << 4   # few distinct values: many ties inside a window
    params = dict(ksize=k, rows=rows, cols=cols, nimg=n)
    (t,) = _dev(ctx, img)
    got = ctx.median_gray(t, k)
    ctx.synchronize()
    got = got.cpu().numpy()
    for i in range(n):
        if not np.array_equal(got[i], oracle.median_gray(img[i], k)):
            return params, "image %d differs" % i
    return params, None


def fuzz_gft(ctx, rng):
    rows, cols = int(rng.integers(3, 140)), int(rng.integers(3, 420))
    nmask, ni = int(rng.integers(1, 7)), int(rng.integers(1, 4))
    cap = int(rng.choice([16, 64, 512, 1024, 2048]))
    shape_mode = rng.random()
    if shape_mode < 0.08:      # more than 1024 problems in one launch: the four-wave selection / descriptor workgroups
        rows, cols, nmask, ni, cap = int(rng.integers(8, 36)), int(rng.integers(20, 90)), int(rng.integers(20, 33)), int(rng.integers(36, 60)), 64
    elif shape_mode < 0.16:    # a large image: many strips and row chunks
        rows, cols = int(rng.integers(140, 320)), int(rng.integers(420, 1500))
    q = float(10 ** rng.uniform(-3, -0.3))
    md = float(rng.choice([0.0, 0.7, 1.0, 1.5, 2.0, 3.0, 5.0, 7.5, 12.0, 25.0, 33.3, 40.0, 60.0]))
    mc = int(rng.choice([0, 1, 7, 50, 300, 1000]))
    params = dict(rows=rows, cols=cols, nmask=nmask, nimg=ni, cap=cap, quality=q, min_distance=md, max_corners=mc)
    kinds = rng.integers(0, 4, ni)
    imgs = np.stack([_image(rng, rows, cols, int(k)) for k in kinds])
    bits = _masks(rng, rows, cols, nmask)[None]
    t_img, t_bits = _dev(ctx, imgs, bits)
    kp, n, status = ctx.detect_gft(t_img, t_bits, ni, nmask, cap, quality=q, min_distance=md, max_corners=mc)
    ctx.synchronize()
    kp, n, status = kp.cpu().numpy(), n.cpu().numpy(), status.cpu().numpy()
    flagged = False
    for i in range(ni):
        eig = oracle.min_eigen(imgs[i])
        for m in range(nmask):
            want, _ = oracle.gft_select(eig, bits[0], m, q, md, mc)
            p = i * nmask + m
            if status[p]:
                flagged = True
                continue
            want = want[:cap]
            if n[p] != len(want) or not np.array_equal(kp[p, :n[p]], want):
                return params, "image %d mask %d: %d corners, oracle %d" % (i, m, n[p], len(want))
    return params, ("flagged" if flagged else None)


def _fast_like(ctx, rng, agast):
    import refflow
    rows, cols = int(rng.integers(8, 120)), int(rng.integers(8, 330))
    if rng.random() < 0.08:
        rows, cols = int(rng.integers(120, 400)), int(rng.integers(330, 1500))
    nmask, ni = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    cap = int(rng.choice([8, 64, 1024, 4096]))
    thr = int(rng.choice([1, 5, 10, 20, 40, 80]))
    params = dict(rows=rows, cols=cols, nmask=nmask, nimg=ni, cap=cap, threshold=thr)
    imgs = np.stack([_image(rng, rows, cols, int(rng.integers(0, 4))) for _ in range(ni)])
    bits = _masks(rng, rows, cols, nmask)[None]
    t_img, t_bits = _dev(ctx, imgs, bits)
    fn = ctx.detect_agast if agast else ctx.detect_fast
    kp, n, status = fn(t_img, t_bits, ni, nmask, cap, threshold=thr)
    ctx.synchronize()
    kp, n, status = kp.cpu().numpy(), n.cpu().numpy(), status.cpu().numpy()
    ref = refflow.agast_keypoints if agast else refflow.fast_keypoints
    for i in range(ni):
        for m in range(nmask):
            want = ref(imgs[i], bits[0], m, thr)
            p = i * nmask + m
            if agast and status[p] & 2:
                continue
            if n[p] != min(len(want), cap) or not np.array_equal(kp[p, :n[p]], want[:cap]):
                return params, "image %d mask %d: %d corners, oracle %d" % (i, m, n[p], len(want))
    return params, None


def fuzz_fast(ctx, rng):
    return _fast_like(ctx, rng, False)


def fuzz_agast(ctx, rng):
    return _fast_like(ctx, rng, True)


def _descriptors(rng, P, Sq, St, nq, nt):
    q = rng.integers(0, 256, (P, Sq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (P, St, 32), dtype=np.uint8)
    mode = int(rng.integers(0, 4))
    for p in range(P):
        if mode == 1 and nq[p] and nt[p]:      # noisy copies + exact duplicates: ties
            m = max(1, min(nq[p], nt[p]) // 2)
            src, dst = rng.integers(0, nq[p], m), rng.integers(0, nt[p], m)
            t[p, dst] = q[p, src] ^ np.packbits(rng.random((m, 32, 8)) < 0.04, axis=-1)[..., 0]
            t[p, dst[:m // 3]] = q[p, src[:m // 3]]
        elif mode == 2:                        # extremes: all zeros / all ones rows
            t[p, ::3] = 0
            q[p, ::5] = 255
        elif mode == 3 and nt[p]:              # every train row the same
            t[p, :] = t[p, 0]
    return q, t


def fuzz_match(ctx, rng):
    P = int(rng.integers(1, 20))
    Sq, St = int(rng.choice([1, 17, 64, 130, 256, 300, 1024, 2100])), int(rng.choice([1, 33, 128, 129, 255, 700, 2100]))
    if P * Sq * St > 3e7:
        P = max(1, int(3e7 // (Sq * St)))
    k = int(rng.integers(1, 3))
    nq = rng.integers(0, Sq + 1, P).astype(np.int32)
    nt = rng.integers(0, St + 1, P).astype(np.int32)
    if rng.random() < 0.5:
        nq[:], nt[:] = Sq, St
    params = dict(P=P, q_stride=Sq, t_stride=St, k=k, nq=nq.tolist()[:8], nt=nt.tolist()[:8])
    q, t = _descriptors(rng, P, Sq, St, nq, nt)
    tq, tt, tnq, tnt = _dev(ctx, q, t, nq, nt)
    keys = ctx.match_hamming(tq, tt, tnq, tnt, k=k)
    ctx.synchronize()
    keys = keys.cpu().numpy()
    for p in range(P):
        want = oracle.match_hamming(q[p, :nq[p]], t[p, :nt[p]], k=k)
        if not np.array_equal(keys[p, :nq[p]], want):
            return params, "problem %d" % p
    return params, None


def fuzz_radius(ctx, rng):
    P = int(rng.integers(1, 8))
    Sq, St = int(rng.choice([1, 40, 200, 333])), int(rng.choice([1, 64, 257, 500]))
    nq = rng.integers(0, Sq + 1, P).astype(np.int32)
    nt = rng.integers(0, St + 1, P).astype(np.int32)
    maxd, cap = float(rng.choice([0, 20, 90, 110, 128, 300])), int(rng.choice([1, 4, 16]))
    params = dict(P=P, q_stride=Sq, t_stride=St, max_distance=maxd, cap=cap)
    q, t = _descriptors(rng, P, Sq, St, nq, nt)
    tq, tt, tnq, tnt = _dev(ctx, q, t, nq, nt)
    keys, cnt = ctx.match_radius(tq, tt, tnq, tnt, maxd, cap)
    ctx.synchronize()
    keys, cnt = keys.cpu().numpy(), cnt.cpu().numpy()
    for p in range(P):
        wk, wc = oracle.match_radius(q[p, :nq[p]], t[p, :nt[p]], maxd, cap)
        if not np.array_equal(cnt[p, :nq[p]], wc):
            return params, "problem %d: counts" % p
        for i in range(nq[p]):
            m = min(int(wc[i]), cap)
            if not np.array_equal(keys[p, i, :m], wk[i, :m]):
                return params, "problem %d query %d" % (p, i)
    return params, None


def fuzz_orb(ctx, rng):
    rows, cols = int(rng.integers(70, 150)), int(rng.integers(70, 400))
    if rng.random() < 0.08:
        rows, cols = int(rng.integers(150, 300)), int(rng.integers(400, 1500))
    nmask, ni = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    nfeat, cap = int(rng.choice([10, 60, 230, 500])), int(rng.choice([64, 512, 1024]))
    params = dict(rows=rows, cols=cols, nmask=nmask, nimg=ni, nfeatures=nfeat, cap=cap)
    imgs = np.stack([_image(rng, rows, cols, int(rng.integers(0, 3))) for _ in range(ni)])
    bits = _masks(rng, rows, cols, nmask)[None]
    t_img, t_bits = _dev(ctx, imgs, bits)
    pyr = ctx.orb_mask_pyramid(t_bits, nmask)
    kp4, resp, n = ctx.detect_orb(t_img, pyr, ni, nmask, nfeat, cap)
    ctx.synchronize()
    kp_d, resp_d, n_d = kp4.cpu().numpy(), resp.cpu().numpy(), n.cpu().numpy()   # (describe_orb_levels compacts kp4 / n in place)
    (pattern,) = _dev(ctx, orb_pattern.orb_pattern())
    desc, _ = ctx.describe_orb_levels(t_img, kp4, n, nmask, pattern)
    ctx.synchronize()
    n2, desc = n.cpu().numpy(), desc.cpu().numpy()
    for i in range(ni):
        want = oracle.orb_detect(imgs[i], bits[0], nmask, nfeat, cap)
        for m in range(nmask):
            p = i * nmask + m
            wkp, wresp = want[m]
            if n_d[p] != len(wkp) or not np.array_equal(kp_d[p, :n_d[p]], wkp) or not np.array_equal(resp_d[p, :n_d[p]], wresp):
                k = min(n_d[p], len(wkp))
                bad = np.nonzero((kp_d[p, :k] != wkp[:k]).any(axis=1) | (resp_d[p, :k] != wresp[:k]))[0]
                first = int(bad[0]) if len(bad) else k
                return params, "image %d mask %d: keypoints (%d, oracle %d); first difference at %d of %d differing: got %s %r, oracle %s %r" % (
                    i, m, n_d[p], len(wkp), first, len(bad), kp_d[p, first].tolist() if first < n_d[p] else None,
                    float(resp_d[p, first]) if first < n_d[p] else None, wkp[first].tolist() if first < len(wkp) else None,
                    float(wresp[first]) if first < len(wkp) else None)
            wd, kept = oracle.orb_describe_levels(imgs[i], wkp)
            if n2[p] != len(kept) or not np.array_equal(desc[p, :n2[p]], wd):
                return params, "image %d mask %d: descriptors (%d, oracle %d)" % (i, m, n2[p], len(kept))
    return params, None


def fuzz_describe(ctx, rng):
    rows, cols = int(rng.integers(64, 200)), int(rng.integers(64, 500))
    nmask, ni = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    cap = int(rng.choice([8, 128, 1024, 2048]))
    if rng.random() < 0.08:
        rows, cols, nmask, ni, cap = 64, int(rng.integers(64, 90)), int(rng.integers(20, 33)), int(rng.integers(36, 56)), 16
    angle = float(rng.choice([orb_pattern.GFT_KEYPOINT_ANGLE, 0.0, 37.5, 90.0, 180.0, 271.3]))
    params = dict(rows=rows, cols=cols, nmask=nmask, nimg=ni, cap=cap, angle=angle)
    imgs = np.stack([_image(rng, rows, cols, int(rng.integers(0, 3))) for _ in range(ni)])
    P = ni * nmask
    kp = np.zeros((P, cap, 2), np.float32)
    n = rng.integers(0, cap + 1, P).astype(np.int32)
    for p in range(P):
        whole = rng.random() < 0.7
        x, y = rng.uniform(0, cols, n[p]), rng.uniform(0, rows, n[p])
        kp[p, :n[p], 0], kp[p, :n[p], 1] = (x.round(), y.round()) if whole else (x, y)
        if n[p] >= 4:      # the border rule's edges
            kp[p, :4] = [[31.0, 31.0], [30.99, 40.0], [cols - 31.0, 40.0], [cols - 31.01, rows - 31.01]]
    pat = orb_pattern.orb_pattern()
    ca, sa = orb_pattern.angle_cos_sin(angle)
    t_img, t_kp, t_n, t_pat = _dev(ctx, imgs, kp, n, pat)
    desc = ctx.describe_orb(t_img, t_kp, t_n, nmask, t_pat, ca, sa)
    ctx.synchronize()
    desc, kp_out, n_out = desc.cpu().numpy(), t_kp.cpu().numpy(), t_n.cpu().numpy()
    blurred = [oracle.gauss7(im) for im in imgs]
    for p in range(P):
        wd, kept = oracle.orb_describe(blurred[p // nmask], kp[p, :n[p]], ca, sa, pat)
        if n_out[p] != len(kept) or not np.array_equal(kp_out[p, :len(kept)], kp[p, kept]):
            return params, "problem %d: border rule (%d kept, oracle %d)" % (p, n_out[p], len(kept))
        if not np.array_equal(desc[p, :len(kept)], wd):
            return params, "problem %d: descriptor bits" % p
    return params, None


def fuzz_unwrap(ctx, rng):
    H, W = int(rng.integers(2, 80)), int(rng.integers(2, 120))
    rows, cols, nf = int(rng.integers(1, 50)), int(rng.integers(1, 150)), int(rng.integers(1, 4))
    ksize = int(rng.choice([0, 3, 5, 11]))
    params = dict(H=H, W=W, rows=rows, cols=cols, nframes=nf, ksize=ksize)
    omni = rng.integers(0, 256, (nf, H, W, 3), dtype=np.uint8)
    span = float(rng.choice([1.0, 1.3, 4.0]))   # maps that stay inside / leave the frame
    mx = (rng.uniform(-0.5 * (span - 1) * W - 5, W * (1 + 0.5 * (span - 1)) + 5, (2, rows, cols))).astype(np.float32)
    my = (rng.uniform(-0.5 * (span - 1) * H - 5, H * (1 + 0.5 * (span - 1)) + 5, (2, rows, cols))).astype(np.float32)
    if rng.random() < 0.5:
        mx = np.round(mx * 64) / 64     # exact 1/64 fractions: the round-half-even cases of the 1/32-pixel grid
        my = np.round(my * 64) / 64
    mx[rng.random(mx.shape) < 0.02] = np.nan
    my[rng.random(my.shape) < 0.02] = np.inf
    masks = ((rng.random((2, H, W)) < rng.uniform(0.3, 1.0)).astype(np.uint8)) * 255
    use_mask = rng.random() < 0.7
    t_omni, t_masks, t_mx, t_my = _dev(ctx, omni, masks, mx.astype(np.float32), my.astype(np.float32))
    tm = t_masks if use_mask else None
    pano = ctx.unwrap(t_omni, tm, t_mx, t_my)
    table = ctx.unwrap_prepare(tm, t_mx, t_my, (H, W))
    pano_t = ctx.unwrap_table(t_omni, table)
    gray = ctx.unwrap_median_gray(t_omni, table, ksize)
    ctx.synchronize()
    pano, pano_t, gray = pano.cpu().numpy(), pano_t.cpu().numpy(), gray.cpu().numpy()
    for v in range(2):
        for f in range(nf):
            want = oracle.unwrap(omni[f], masks[v] if use_mask else None, mx[v], my[v])
            if not np.array_equal(pano[v, f], want):
                return params, "view %d frame %d: map form" % (v, f)
            if not np.array_equal(pano_t[v, f], want):
                return params, "view %d frame %d: table form" % (v, f)
            if not np.array_equal(gray[v * nf + f], oracle.median_gray(want, ksize)):
                return params, "view %d frame %d: fused unwrap + median + gray" % (v, f)
    return params, None


def fuzz_ransac(ctx, rng):
    import synth
    P = int(rng.integers(1, 6))
    S = int(rng.choice([8, 40, 130, 700]))
    noncentral = bool(rng.integers(0, 2))
    solver = str(rng.choice(["P3P", "GP3P", "EPNP", "TWOPT"] if True else []))
    if solver == "GP3P":
        noncentral = True
    if solver in ("EPNP", "TWOPT"):
        noncentral = False
    max_iter = int(rng.choice([1, 17, 64, 300]))
    adaptive = bool(rng.integers(0, 2))
    thr = float(rng.choice([synth.THR_5DEG, 1.0 - np.cos(np.deg2rad(0.5)), 1e-9]))
    seed = int(rng.integers(0, 2 ** 31))
    params = dict(P=P, S=S, noncentral=noncentral, solver=solver, max_iter=max_iter, adaptive=adaptive, thr=thr, seed=seed)
    problems = []
    for b in range(P):
        n = int(rng.integers(0, S + 1)) if rng.random() < 0.7 else S
        pr = synth.make_abs_pose_problem(rng, max(n, 1), inlier_frac=float(rng.choice([0.0, 0.2, 0.6, 1.0])),
                                         noise_deg=float(rng.choice([0.0, 0.2, 2.0])), noncentral=noncentral)
        if n == 0:
            pr = dict(pr, f=pr["f"][:0], p=pr["p"][:0], cam=None if pr["cam"] is None else pr["cam"][:0])
        if solver == "TWOPT":     # the identity-rotation solver: bearings of a pure translation
            pr["f"] = pr["f"]
        problems.append(pr)
    f = np.zeros((P, S, 3)); pp = np.zeros((P, S, 3)); cam = np.zeros((P, S), np.int32); n_arr = np.zeros(P, np.int32)
    for b, pr in enumerate(problems):
        k = pr["f"].shape[0]
        n_arr[b] = k
        f[b, :k], pp[b, :k] = pr["f"], pr["p"]
        if noncentral:
            cam[b, :k] = pr["cam"]
    tf, tp, tcam, tn = _dev(ctx, f, pp, cam, n_arr)
    kw = {}
    if noncentral:
        off, rot = _dev(ctx, problems[0]["cam_off"], problems[0]["cam_rot"])
        kw = dict(cam=tcam, cam_off=off, cam_rot=rot, cam_rot_identity=bool(rng.integers(0, 2)))
    flags = dict(epnp=solver == "EPNP", gp3p=solver == "GP3P", twopt=solver == "TWOPT")
    out = ctx.ransac_abs_pose(tf, tp, tn, thr, max_iter, seed=seed, adaptive=adaptive, want_counts=True, **flags, **kw)
    T0 = out["T"].clone()
    kw2 = {k: v for k, v in kw.items() if k != "cam_rot_identity"}
    Tr, cost, iters = ctx.refine_abs_pose(tf, tp, tn, out["T"].clone(), idx=out["idx"], m=out["n_inliers"], **kw2)
    ctx.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    T0, Tr, cost, iters = T0.cpu().numpy(), Tr.cpu().numpy(), cost.cpu().numpy(), iters.cpu().numpy()
    for b, pr in enumerate(problems):
        okw = dict(cam=pr["cam"], cam_off=pr["cam_off"], cam_rot=pr["cam_rot"]) if noncentral else {}
        want = oracle.ransac_abs_pose(pr["f"], pr["p"], thr, max_iter, seed=seed + b, adaptive=adaptive, want_counts=True, **flags, **okw)
        k, used = n_arr[b], want["iters_used"]
        if got["info"][b, 1] != used or got["info"][b, 0] != want["best_iter"] or got["info"][b, 2] != want["status"]:
            return params, "problem %d: iterations / winner / status (%s, oracle %d %d %d)" % (b, got["info"][b, :3].tolist(), want["best_iter"], used, want["status"])
        if not np.array_equal(got["counts"][b, :used], want["counts"][:used]):
            return params, "problem %d: hypothesis counts" % b
        if got["n_inliers"][b] != want["n_inliers"] or not np.array_equal(got["mask"][b, :k].astype(bool), want["mask"]):
            return params, "problem %d: inlier mask" % b
        if not np.array_equal(got["T"][b], want["T"]):
            return params, "problem %d: pose bits" % b
        if want["status"] == 0 and want["n_inliers"] >= 3:
            idx = np.nonzero(want["mask"])[0].astype(np.int32)
            wT, wcost, wit = oracle.refine_abs_pose(pr["f"], pr["p"], T0[b], idx=idx, **okw)
            if not np.array_equal(Tr[b], wT) or iters[b] != wit:
                return params, "problem %d: refined pose bits / LM iterations (%d, oracle %d)" % (b, iters[b], wit)
    return params, None


def fuzz_relpose(ctx, rng):
    from test_oracle_relpose import _two_views
    P, S = int(rng.integers(1, 5)), int(rng.choice([16, 130, 600]))
    algorithm = int(rng.choice([5, 7, 8]))
    max_iter, adaptive, seed = int(rng.choice([1, 33, 200])), bool(rng.integers(0, 2)), int(rng.integers(0, 2 ** 31))
    thr = float(rng.choice([2.0 * (1.0 - np.cos(np.deg2rad(1.0))), 2.0 * (1.0 - np.cos(np.deg2rad(0.1))), 1e-12]))
    params = dict(P=P, S=S, algorithm=algorithm, max_iter=max_iter, adaptive=adaptive, seed=seed, thr=thr)
    problems = []
    for b in range(P):
        n = int(rng.integers(0, S + 1))
        a, c, _, _, _ = _two_views(rng, max(n, 1), noise_deg=float(rng.choice([0.0, 0.05, 0.5])), outlier_frac=float(rng.choice([0.0, 0.3, 0.9])))
        problems.append((a[:n], c[:n]))
    f1 = np.zeros((P, S, 3)); f2 = np.zeros((P, S, 3)); n_arr = np.zeros(P, np.int32)
    for b, (a, c) in enumerate(problems):
        n_arr[b] = a.shape[0]
        f1[b, :n_arr[b]], f2[b, :n_arr[b]] = a, c
    t1, t2, tn = _dev(ctx, f1, f2, n_arr)
    out = ctx.ransac_rel_pose(t1, t2, tn, thr, max_iter, algorithm=algorithm, seed=seed, adaptive=adaptive, want_counts=True)
    ctx.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    for b, (a, c) in enumerate(problems):
        want = oracle.ransac_rel_pose(a, c, thr, max_iter, seed=seed + b, adaptive=adaptive, want_counts=True, algorithm=algorithm)
        used = want["iters_used"]
        if got["info"][b, 1] != used or got["info"][b, 0] != want["best_iter"] or got["info"][b, 2] != want["status"]:
            return params, "problem %d: iterations / winner / status" % b
        if not np.array_equal(got["counts"][b, :used], want["counts"][:used]):
            return params, "problem %d: hypothesis counts" % b
        if got["n_inliers"][b] != want["n_inliers"] or not np.array_equal(got["mask"][b, :n_arr[b]].astype(bool), want["mask"]):
            return params, "problem %d: inlier mask" % b
        if not np.array_equal(got["T"][b].view(np.uint64), want["T"].view(np.uint64)):
            return params, "problem %d: pose bits" % b
    return params, None


def fuzz_l2sort(ctx, rng):
    """float-descriptor matching (sosvo_match_l2) and the stable sort of match keys"""
    P, Sq, St, k = int(rng.integers(1, 6)), int(rng.choice([1, 50, 257])), int(rng.choice([1, 64, 300])), int(rng.integers(1, 3))
    D = int(rng.choice([32, 64, 128]))
    params = dict(P=P, q_stride=Sq, t_stride=St, k=k, dim=D)
    q = rng.integers(0, 8, (P, Sq, D)).astype(np.float32)   # small integers: exact sums, many ties
    t = rng.integers(0, 8, (P, St, D)).astype(np.float32)
    nq, nt = rng.integers(0, Sq + 1, P).astype(np.int32), rng.integers(0, St + 1, P).astype(np.int32)
    tq, tt, tnq, tnt = _dev(ctx, q, t, nq, nt)
    res = ctx.match_l2(tq, tt, tnq, tnt, k=k)
    # Hamming keys for the sort: random keys with many equal distances
    hk = (rng.integers(0, 6, (P, Sq, 1)).astype(np.uint32) << np.uint32(20)) | rng.integers(0, 1000, (P, Sq, 1)).astype(np.uint32)
    (thk,) = _dev(ctx, hk)
    order = ctx.sort_matches(thk, tnq)
    ctx.synchronize()
    res = res.cpu().numpy().view(np.uint64)
    order = order.cpu().numpy()
    for p in range(P):
        if nq[p] and not np.array_equal(res[p, :nq[p]], oracle.match_l2(q[p, :nq[p]], t[p, :nt[p]].reshape(-1, D), k=k)):
            return params, "problem %d: L2 matches" % p
        wo = oracle.sort_matches(hk[p, :nq[p]])
        if not np.array_equal(order[p, :nq[p]], wo):
            return params, "problem %d: sort order" % p
    return params, None


def fuzz_pipeline(ctx, rng):
    """Everything after detection (bucket matching, gates, triangulation, frame-to-frame matching, RANSAC, LM) on image-free
    frames against the reference's control flow on the oracle (tests/refflow.py)."""
    import refflow
    import synth
    from vo_single_camera_sos_amd.pipeline import FramePairPipeline, RigConfig
    B, NM = int(rng.integers(1, 5)), int(rng.choice([1, 4, 12]))
    cap = int(rng.choice([16, 64, 192]))
    frame_cap = int(rng.choice([256, 2048]))
    max_iter, adaptive = int(rng.choice([10, 100, 400])), bool(rng.integers(0, 2))
    solver = str(rng.choice(["P3P", "GP3P"]))
    rig_kw = dict(pano_top=synth.PANO_C2, pano_bot=synth.PANO_C2, F_top=synth.F_TOP, F_bot=synth.F_BOT,
                  min_range=float(rng.choice([0.0, 500.0, 2000.0])), max_range=float(rng.choice([3000.0, 7000.0, 1e9])),
                  stereo_min_disp=float(rng.choice([0.0, 1.0, 3.0])), stereo_max_hdiff=float(rng.choice([0.5, 2.5, 1e6])),
                  f2f_max_hdiff=float(rng.choice([-1.0, 10.0, 75.0])), pct_good_matches=float(rng.choice([1.0, 0.8, 0.3])))
    seed = int(rng.integers(0, 2 ** 31))
    params = dict(B=B, nmask=NM, bucket_cap=cap, frame_cap=frame_cap, max_iter=max_iter, adaptive=adaptive, solver=solver, seed=seed,
                  **{k: v for k, v in rig_kw.items() if isinstance(v, float)})
    frames = []
    for i in range(B):
        P, desc = synth.make_scene(rng, int(rng.choice([0, 30, 400, 1500])))
        R, t = synth.random_pose(rng)
        kw = dict(nmask=NM, cap=cap, flip_prob=float(rng.choice([0.0, 0.04, 0.2])), distractors=float(rng.choice([0.0, 0.15, 1.0])))
        frames.append(synth.observe_frame(rng, P, desc, np.eye(3), np.zeros(3), **kw))
        frames.append(synth.observe_frame(rng, P, desc, R, t, **kw))
    pipe = FramePairPipeline(ctx, RigConfig(**rig_kw), B, nmask=NM, bucket_cap=cap, frame_cap=frame_cap, max_iter=max_iter, adaptive=adaptive,
                             seed=seed, ransac_solver=solver)
    pipe.load_keypoints(synth.pack_buckets(frames, NM, cap))
    pipe.step()
    rec = pipe.results()
    ctx.synchronize()
    rec = rec.cpu().numpy()
    mask = pipe.ransac["mask"].cpu().numpy()
    M = pipe.frames["M"].cpu().numpy()
    rp = refflow.RigParams(**rig_kw)
    flagged = False
    for i in range(B):
        ref = refflow.stereo_frame(rp, *[frames[2 * i][k] for k in ("kp_top", "kp_bot", "desc_top", "desc_bot")])
        cur = refflow.stereo_frame(rp, *[frames[2 * i + 1][k] for k in ("kp_top", "kp_bot", "desc_top", "desc_bot")])
        if len(ref["X"]) > frame_cap or len(cur["X"]) > frame_cap:
            flagged = True     # more stereo points than the frame store holds: the first frame_cap in order are kept (sosvo.h)
            ref, cur = [{k: (v[:frame_cap] if k != "n_cand" else v) for k, v in fr.items()} for fr in (ref, cur)]
        if M[2 * i] != len(ref["X"]) or M[2 * i + 1] != len(cur["X"]):
            return params, "pair %d: stereo points (%d %d, oracle %d %d)" % (i, M[2 * i], M[2 * i + 1], len(ref["X"]), len(cur["X"]))
        w = refflow.track_pair(rp, ref, cur, pipe.thr, max_iter, seed=seed + i, adaptive=adaptive, gp3p=solver == "GP3P")
        n = len(w["corr"]["cam"])
        if rec[i, 13] != n:
            return params, "pair %d: correspondences (%d, oracle %d)" % (i, rec[i, 13], n)
        if rec[i, 14] != w["ransac"]["status"] or rec[i, 15] != w["ransac"]["best_iter"] or rec[i, 12] != w["ransac"]["n_inliers"]:
            return params, "pair %d: RANSAC status / winner / inliers" % i
        if not np.array_equal(mask[i, :n].astype(bool), w["ransac"]["mask"]):
            return params, "pair %d: inlier mask" % i
        if not np.array_equal(rec[i, :12].reshape(3, 4), w["T"]):
            return params, "pair %d: pose bits" % i
    return params, ("flagged" if flagged else None)


def fuzz_rgbd(ctx, rng):
    """The RGB-D (perspective) path behind ONE C-ABI call (sosvo_rgbd_pair_batch) on small random frames -- textures shifted
    between the frames of a pair, depth maps with holes / NaN / out-of-range values, both depth conventions, all solvers,
    an optional median window -- against the reference's control flow on the oracle (tests/refflow.py)."""
    import refflow
    from vo_single_camera_sos_amd.pipeline import RGBDCamConfig, RGBDPairBatch
    B = int(rng.integers(1, 4))
    rows, cols = int(rng.integers(80, 200)), int(rng.integers(100, 320))
    nfeat = int(rng.choice([50, 400, 2000]))
    algo = str(rng.choice(["EPNP", "KNEIP", "GP3P", "TWOPT"]))
    depth_is_Z = bool(rng.integers(0, 2))
    ksize = int(rng.choice([0, 0, 3, 5]))
    max_iter, adaptive, seed = int(rng.choice([20, 200])), bool(rng.integers(0, 2)), int(rng.integers(0, 2 ** 31))
    thr = float(rng.choice([1.0 - np.cos(np.deg2rad(5.0)), 1.0 - np.cos(np.deg2rad(0.5))]))
    fx, fy, cx, cy = 0.8 * cols, 0.8 * cols, 0.5 * cols - 0.5, 0.5 * rows - 0.5
    params = dict(B=B, rows=rows, cols=cols, nfeat=nfeat, algo=algo, depth_is_Z=depth_is_Z, ksize=ksize, max_iter=max_iter,
                  adaptive=adaptive, seed=seed, thr=thr)
    bgr, depth = [], []
    for i in range(B):
        base = np.stack([_image(rng, rows + 8, cols + 8, int(rng.integers(0, 2))) for _ in range(3)], axis=-1)
        dz = rng.uniform(0.5, 9.0, (rows + 8, cols + 8)).astype(np.float32)     # metres * 1000 below; some beyond the range gates
        dz = np.kron(dz[::8, ::8], np.ones((8, 8), np.float32))[:rows + 8, :cols + 8]
        sx, sy = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        for (ox, oy) in ((0, 0), (sx, sy)):
            im = np.ascontiguousarray(base[oy:oy + rows, ox:ox + cols])
            d = (1000.0 * dz[oy:oy + rows, ox:ox + cols]).astype(np.float32)
            hole = rng.random((rows, cols)) < float(rng.choice([0.0, 0.1, 0.9]))
            d[hole] = float(rng.choice([0.0, np.nan]))
            bgr.append(im)
            depth.append(d)
    bgr, depth = np.stack(bgr), np.stack(depth)
    cam = RGBDCamConfig(fx=fx, fy=fy, center_x=cx, center_y=cy, depth_is_Z=depth_is_Z, min_range=0.8, max_range=7.0)
    one = RGBDPairBatch(ctx, cam, B, image_shape=(rows, cols), num_of_features=nfeat, max_iter=max_iter, adaptive=adaptive, seed=seed,
                        thr=thr, pose_est_algorithm=algo, median_win_size=ksize)
    one.load_frames(bgr, depth)
    rec = one.step()
    ctx.synchronize()
    rec = rec.cpu().numpy()
    ca, sa = orb_pattern.angle_cos_sin(orb_pattern.GFT_KEYPOINT_ANGLE)
    rc = refflow.RGBDParams(fx, fy, cx, cy, cam.focal_length_m, depth_is_Z, 0.8, 7.0, cam.f2f_max_hdiff, 1.0)
    frames = [refflow.rgbd_frame(rc, bgr[f], depth[f], nfeat, orb_pattern.orb_pattern(), ca, sa, median_ksize=ksize) for f in range(2 * B)]
    for i in range(B):
        w = refflow.track_pair_rgbd(rc, frames[2 * i], frames[2 * i + 1], thr, max_iter, seed=seed + i, adaptive=adaptive,
                                    epnp=(algo == "EPNP"), gp3p=(algo == "GP3P"), twopt=(algo == "TWOPT"))
        n = len(w["corr"]["q"])
        if rec[i, 13] != n:
            return params, "pair %d: correspondences (%d, oracle %d)" % (i, rec[i, 13], n)
        if rec[i, 14] != w["ransac"]["status"] or rec[i, 15] != w["ransac"]["best_iter"] or rec[i, 12] != w["ransac"]["n_inliers"]:
            return params, "pair %d: RANSAC status / winner / inliers (%s, oracle %d %d %d)" % (
                i, rec[i, 12:16].tolist(), w["ransac"]["n_inliers"], w["ransac"]["status"], w["ransac"]["best_iter"])
        if not np.array_equal(rec[i, :12].reshape(3, 4), w["T"]):
            return params, "pair %d: pose bits" % i
    return params, None


STAGES = {"median": fuzz_median, "gft": fuzz_gft, "fast": fuzz_fast, "agast": fuzz_agast, "match": fuzz_match,
          "radius": fuzz_radius, "orb": fuzz_orb, "unwrap": fuzz_unwrap, "ransac": fuzz_ransac,
          "describe": fuzz_describe, "relpose": fuzz_relpose, "l2sort": fuzz_l2sort,
          "pipeline": fuzz_pipeline, "rgbd": fuzz_rgbd}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--stages", default=",".join(STAGES))
    ap.add_argument("--first-case", type=int, default=0)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases per stage (0 = by time)")
    args = ap.parse_args()
    oracle.build()
    ctx = Context(0)
    stages = [s for s in args.stages.split(",") if s]
    t_end = time.time() + 60 * args.minutes
    done = {s: 0 for s in stages}
    flagged = {s: 0 for s in stages}
    refused = {}
    case = args.first_case
    t_print = time.time()
    while time.time() < t_end and (args.cases == 0 or case < args.first_case + args.cases):
        for s in stages:
            rng = np.random.default_rng([args.seed, list(STAGES).index(s), case])
            try:
                params, err = STAGES[s](ctx, rng)
            except SosvoError as e:     # an argument the entry point refuses (documented ranges): shown once per message
                msg = str(e).split(":")[-1].strip()
                if msg not in refused:
                    refused[msg] = (s, case)
                    print("refused (stage %s case %d): %s" % (s, case, e), flush=True)
                continue
            if err == "flagged":
                flagged[s] += 1
            elif err:
                print("MISMATCH stage %s case %d (seed %d): %s\n  %s" % (s, case, args.seed, err, params), flush=True)
                return 1
            done[s] += 1
        case += 1
        if time.time() - t_print > 20:
            t_print = time.time()
            print("cases %s flagged %s" % (done, flagged), flush=True)
    print("fuzz_parity: all identical.  cases %s flagged %s seed %d first-case %d" % (done, flagged, args.seed, args.first_case), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
