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


def pose_error(T, R, t):
    """(rotation angle error [rad], relative translation error)."""
    dR = T[:, :3].T @ R
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    return ang, np.linalg.norm(T[:, 3] - t) / max(np.linalg.norm(t), 1e-12)
