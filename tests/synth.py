"""Synthetic, seeded inputs for the image-free stages: 3-D points around the rig,
a known SE(3) step, bearings in the top/bottom mirror frames, planted outliers and angular
noise.  Own code; nothing here comes from the reference."""
import numpy as np

THR_5DEG = 1.0 - np.cos(np.deg2rad(5.0))  # pose_est_tools.py:675-676 -> 0.003805301908254455
F_TOP = np.array([0.0, 0.0, 150.0])  # mm, foci of the validated synthetic GUMS
F_BOT = np.array([0.0, 0.0, 50.0])


def rot_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def random_pose(rng, max_t=100.0, max_deg=5.0):
    """Pose of the current viewpoint in the reference frame: X_ref = R x_cur + t."""
    axis = rng.normal(size=3)
    ang = np.deg2rad(rng.uniform(0.2, max_deg))
    t = rng.normal(size=3)
    t = t / np.linalg.norm(t) * rng.uniform(0.1 * max_t, max_t)
    return rot_from_axis_angle(axis, ang), t


def perturb_bearings(rng, f, sigma_deg):
    if sigma_deg <= 0:
        return f
    noise = rng.normal(size=f.shape) * np.deg2rad(sigma_deg)
    g = f + noise - (np.sum(noise * f, axis=1, keepdims=True)) * f
    return g / np.linalg.norm(g, axis=1, keepdims=True)


def make_abs_pose_problem(rng, n, inlier_frac=0.35, noise_deg=0.2, noncentral=True, n_top=None,
                          shell=(800.0, 6000.0)):
    """-> dict(f [n,3], p [n,3], cam [n] i32 or None, cam_off, cam_rot, R, t, is_inlier)."""
    R, t = random_pose(rng)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = d * rng.uniform(shell[0], shell[1], size=(n, 1))  # keyframe 3-D points, frame [C], mm
    if noncentral:
        if n_top is None:
            n_top = n // 2
        cam = np.concatenate([np.zeros(n_top, np.int32), np.ones(n - n_top, np.int32)])
        cam_off = np.stack([F_TOP, F_BOT])
        cam_rot = np.stack([np.eye(3), np.eye(3)])
        o = cam_off[cam]
    else:
        cam, cam_off, cam_rot = None, None, None
        o = np.zeros((n, 3))
    v = (p - t) @ R - o  # R^T (p - t) - o_c, rows
    f = v / np.linalg.norm(v, axis=1, keepdims=True)
    f = perturb_bearings(rng, f, noise_deg)
    is_in = rng.random(n) < inlier_frac
    if inlier_frac >= 1.0:
        is_in[:] = True
    bad = rng.normal(size=(n, 3))
    bad /= np.linalg.norm(bad, axis=1, keepdims=True)
    f = np.where(is_in[:, None], f, bad)
    return dict(f=np.ascontiguousarray(f), p=np.ascontiguousarray(p), cam=cam, cam_off=cam_off,
                cam_rot=cam_rot, R=R, t=t, is_inlier=is_in)


# ---- image-free frame generator: keypoints + descriptors on the two panoramas -----------------
PANO_C2 = (1200.0, 122.0, 0.005235987755982988, 0.2681943150929834)  # cols, rows, pixel_size, h_max (App. D)
ELEV_TOP = (-0.35290654146694395, 0.2618998070797145)
ELEV_BOT = (-0.3487218912619687, 0.26202807633801434)


def project_to_pano(P, F, pano, elev):
    """3-D points [n,3] (frame [C], identity mirror rotation) -> pano pixel (u, v) and validity.
    Inverse of the closed form of panorama.py:635-641 / :616-622."""
    cols, rows, px, hmax = pano
    d = P - F
    psi = np.mod(np.arctan2(d[:, 1], d[:, 0]), 2 * np.pi)
    theta = np.arctan2(d[:, 2], np.hypot(d[:, 0], d[:, 1]))
    u = np.mod((2 * np.pi - psi) / px, cols)
    v = (hmax - np.tan(theta)) / px
    ok = (v >= 0) & (v < rows) & (theta >= elev[0]) & (theta <= elev[1]) & (u >= 0) & (u < cols)
    return u, v, ok


def make_scene(rng, n_points):
    """World points (mm, frame of the first viewpoint) inside the rig's common field of view."""
    az = rng.uniform(0, 2 * np.pi, n_points)
    el = np.deg2rad(rng.uniform(-17.0, 12.0, n_points))
    rad = rng.uniform(900.0, 5500.0, n_points)
    base = np.array([0.0, 0.0, 100.0])
    P = base + np.stack([rad * np.cos(el) * np.cos(az), rad * np.cos(el) * np.sin(az), rad * np.sin(el)], axis=1)
    desc = rng.integers(0, 256, (n_points, 32), dtype=np.uint8)
    return P, desc


def observe_frame(rng, P, desc, R, t, nmask=12, cap=256, flip_prob=0.04, px_noise=0.15, distractors=0.15,
                  pano=PANO_C2):
    """Keypoints/descriptors one frame would deliver per azimuthal bucket (camera_models.py:1730 loop):
    lists over buckets of kp_top/kp_bot [n,2] f32 and desc_top/desc_bot [n,32] u8.  The viewpoint
    pose is X_world = R x + t."""
    Pc = (P - t) @ R  # R^T (P - t)
    out = {}
    cols = pano[0]
    for name, F, elev in (("top", F_TOP, ELEV_TOP), ("bot", F_BOT, ELEV_BOT)):
        u, v, ok = project_to_pano(Pc, F, pano, elev)
        idx = np.nonzero(ok)[0]
        uu = u[idx] + rng.normal(0, px_noise, idx.size)
        vv = v[idx] + rng.normal(0, px_noise, idx.size)
        flips = rng.random((idx.size, 32, 8)) < flip_prob
        dd = desc[idx] ^ np.packbits(flips, axis=-1)[..., 0]
        nd = int(distractors * idx.size)
        uu = np.concatenate([uu, rng.uniform(0, cols, nd)])
        vv = np.concatenate([vv, rng.uniform(2, pano[1] - 2, nd)])
        dd = np.concatenate([dd, rng.integers(0, 256, (nd, 32), dtype=np.uint8)])
        keep = (uu >= 0) & (uu < cols) & (vv >= 0) & (vv < pano[1])
        uu, vv, dd = uu[keep], vv[keep], dd[keep]
        perm = rng.permutation(uu.size)
        uu, vv, dd = uu[perm], vv[perm], dd[perm]
        bucket = np.minimum((uu / (cols / nmask)).astype(np.int64), nmask - 1)
        kps, des = [], []
        for m in range(nmask):
            sel = np.nonzero(bucket == m)[0][:cap]
            kps.append(np.stack([uu[sel], vv[sel]], axis=1).astype(np.float32))
            des.append(np.ascontiguousarray(dd[sel]))
        out["kp_" + name], out["desc_" + name] = kps, des
    return out


def pack_buckets(frames, nmask, cap):
    """List of observe_frame dicts -> fixed-capacity arrays for the C ABI:
    kp_top/kp_bot [F*NM, cap, 2] f32, desc_top/desc_bot [F*NM, cap, 32] u8, n_top/n_bot [F*NM] i32."""
    F = len(frames)
    out = dict(kp_top=np.zeros((F * nmask, cap, 2), np.float32), kp_bot=np.zeros((F * nmask, cap, 2), np.float32),
               desc_top=np.zeros((F * nmask, cap, 32), np.uint8), desc_bot=np.zeros((F * nmask, cap, 32), np.uint8),
               n_top=np.zeros(F * nmask, np.int32), n_bot=np.zeros(F * nmask, np.int32))
    for fi, fr in enumerate(frames):
        for m in range(nmask):
            p = fi * nmask + m
            for name in ("top", "bot"):
                k = fr["kp_" + name][m]
                n = len(k)
                out["n_" + name][p] = n
                out["kp_" + name][p, :n] = k
                out["desc_" + name][p, :n] = fr["desc_" + name][m]
    return out


def pose_error(T, R, t):
    """(rotation angle error [rad], relative translation error)."""
    dR = T[:, :3].T @ R
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    return ang, np.linalg.norm(T[:, 3] - t) / max(np.linalg.norm(t), 1e-12)
