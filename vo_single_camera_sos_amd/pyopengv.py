"""Drop-in for the `pyopengv` functions the reference calls, numpy in / numpy out, computed by
libsosvo on the GPU (no CPU fallback: importing works anywhere, calling needs the MI355X library).

    absolute_pose_noncentral_ransac              omnistereo/pose_est_tools.py:785
    absolute_pose_noncentral_optimize_nonlinear  omnistereo/pose_est_tools.py:830
    absolute_pose_ransac                         omnistereo/pose_est_tools.py:116, :915
    absolute_pose_optimize_nonlinear             omnistereo/pose_est_tools.py:937
    triangulation_triangulate2                   omnistereo/pose_est_tools.py:359, :163
    relative_pose_ransac                         omnistereo/pose_est_tools.py:78  (2D-2D; not reached by the VO drivers)

Conventions are OpenGV's as the reference uses them: T = [R | t] (3x4) is the pose of the current viewpoint in
the frame the points are given in (points map by R^T (p - t), pose_est_tools.py:155-160, :177); `threshold`
is 1 - cos(angle); inlier indices come back ascending (relied on at pose_est_tools.py:787-806).

Differences from OpenGV, by design (deviation table):
  * minimal solvers are restatements of the published algorithms with own numerics: the non-central call draws 4
    correspondences across all cameras and solves the generalised P3P (as OpenGV does), the central names map to
    Kneip's P3P ("KNEIP"), the depth formulation through the generalised solver ("GAO", "GP3P"), EPnP on 6-point samples
    ("EPNP"), the 2-point translation solver with the binding's identity rotation prior ("TWOPT"); the relative pose
    offers the five-point ("STEWENIUS", "NISTER": Nister's algorithm), seven-point and eight-point solvers;
  * sampling is a counter-based generator: results are a pure function of (inputs, seed).  `set_seed` fixes the
    seed of the next call; every call advances it by one (OpenGV seeds from the clock)."""
import numpy as np

_seed = [0]
LM_MAX_ITERATIONS = 30


def set_seed(seed):
    _seed[0] = int(seed)


def _next_seed():
    s = _seed[0]
    _seed[0] = s + 1
    return s


def _ctx():
    from .runtime import default_context
    return default_context()


def _dev(ctx, a, dtype, shape=None):
    import torch
    a = np.ascontiguousarray(np.asarray(a), dtype=dtype)
    if shape is not None:
        a = a.reshape(shape)
    return torch.from_numpy(a).to(ctx.device)


def _problem(ctx, b, p, cam_idx=None):
    b = np.asarray(b, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    if b.ndim != 2 or b.shape[1] != 3 or p.shape != b.shape:
        raise ValueError("bearings and points must both be [n, 3]; got %s and %s" % (b.shape, p.shape))
    n = b.shape[0]
    S = max(n, 1)
    f_t = _dev(ctx, b if n else np.zeros((1, 3)), np.float64, (1, S, 3))
    p_t = _dev(ctx, p if n else np.zeros((1, 3)), np.float64, (1, S, 3))
    n_t = _dev(ctx, [n], np.int32)
    cam_t = None
    if cam_idx is not None:
        c = np.asarray(cam_idx).reshape(-1)
        if c.shape[0] != n:
            raise ValueError("cam_idx must have one entry per correspondence")
        cam_t = _dev(ctx, np.rint(c) if n else np.zeros(1), np.int32, (1, S))  # the reference passes floats (:767)
    return f_t, p_t, n_t, cam_t, n


def _rig(ctx, cam_offsets, cam_rotations):
    off = np.asarray(cam_offsets, dtype=np.float64).reshape(-1, 3)
    rot = np.asarray(cam_rotations, dtype=np.float64).reshape(-1, 3, 3)
    if off.shape[0] != rot.shape[0]:
        raise ValueError("cam_offsets and cam_rotations disagree on the number of cameras")
    return _dev(ctx, off, np.float64), _dev(ctx, rot, np.float64)


def _ransac(b, p, threshold, max_iterations, cam_idx=None, cam_offsets=None, cam_rotations=None, epnp=False, gp3p=False,
            twopt=False):
    ctx = _ctx()
    f_t, p_t, n_t, cam_t, n = _problem(ctx, b, p, cam_idx)
    kw = {}
    if cam_t is not None:
        off_t, rot_t = _rig(ctx, cam_offsets, cam_rotations)
        kw = dict(cam=cam_t, cam_off=off_t, cam_rot=rot_t)
    out = ctx.ransac_abs_pose(f_t, p_t, n_t, float(threshold), int(max_iterations), seed=_next_seed(), adaptive=True,
                              epnp=epnp, gp3p=gp3p, twopt=twopt, **kw)
    ctx.synchronize()
    k = int(out["n_inliers"][0].item())
    T = out["T"][0].cpu().numpy()
    inliers = out["idx"][0, :k].cpu().numpy().astype(np.int64)
    return T, inliers


def _refine(b, p, t, R, cam_idx=None, cam_offsets=None, cam_rotations=None):
    import torch
    ctx = _ctx()
    f_t, p_t, n_t, cam_t, n = _problem(ctx, b, p, cam_idx)
    kw = {}
    if cam_t is not None:
        off_t, rot_t = _rig(ctx, cam_offsets, cam_rotations)
        kw = dict(cam=cam_t, cam_off=off_t, cam_rot=rot_t)
    T0 = np.hstack([np.asarray(R, dtype=np.float64).reshape(3, 3), np.asarray(t, dtype=np.float64).reshape(3, 1)])
    T = torch.from_numpy(np.ascontiguousarray(T0[None])).to(ctx.device)
    ctx.refine_abs_pose(f_t, p_t, n_t, T, max_lm_iter=LM_MAX_ITERATIONS, **kw)
    ctx.synchronize()
    return T[0].cpu().numpy()


def absolute_pose_noncentral_ransac(b, cam_idx, p, cam_offsets, cam_rotations, threshold, max_iterations):
    """-> (T [3,4], inlier indices int64 [k] ascending).  Hypotheses from the generalised P3P on samples across all
    cameras, as OpenGV does ("The non-central case will ALWAYS use GP3P", pose_est_tools.py:696)."""
    return _ransac(b, p, threshold, max_iterations, cam_idx, cam_offsets, cam_rotations, gp3p=True)


def absolute_pose_noncentral_optimize_nonlinear(b, cam_idx, p, cam_offsets, cam_rotations, t, R):
    """-> T [3,4]: Levenberg-Marquardt over (t, Cayley(R)) on all given correspondences."""
    return _refine(b, p, t, R, cam_idx, cam_offsets, cam_rotations)


def absolute_pose_ransac(b, p, algo_name, threshold, max_iterations):
    """-> (T [3,4], inlier indices int64 [k] ascending)."""
    name = str(algo_name).upper()
    if name not in ("TWOPT", "KNEIP", "GAO", "EPNP", "GP3P"):
        raise ValueError("unknown algorithm %r" % algo_name)
    # "EPNP": 6-point samples solved by EPnP, as in OpenGV.  "KNEIP": Kneip's closed-form P3P (camera pose from two
    # angles) + a 4th point.  "GAO" and "GP3P": the three-point problem posed on the three DEPTHS (the law-of-cosines
    # system Gao et al. solve), here through the generalised solver on a one-camera rig, + a 4th point.  "TWOPT": 2-point
    # samples, translation only, rotation = the binding's identity prior.
    return _ransac(b, p, threshold, max_iterations, epnp=(name == "EPNP"), gp3p=(name in ("GP3P", "GAO")),
                   twopt=(name == "TWOPT"))


def absolute_pose_optimize_nonlinear(b, p, t, R):
    """-> T [3,4]."""
    return _refine(b, p, t, R)


REL_ALGORITHMS = {"STEWENIUS": 5, "NISTER": 5, "SEVENPT": 7, "EIGHTPT": 8}   # both five-point names: Nister's algorithm


def relative_pose_ransac(b1, b2, algo_name, threshold, max_iterations):
    """-> (T [3,4] = [R | t], the pose of viewpoint 2 in frame 1 with |t| = 1, inlier indices int64 [k] ascending)."""
    name = str(algo_name).upper()
    if name not in REL_ALGORITHMS:
        raise ValueError("unknown relative-pose algorithm %r (have: %s)" % (algo_name, ", ".join(sorted(REL_ALGORITHMS))))
    ctx = _ctx()
    f1_t, f2_t, n_t, _, n = _problem(ctx, b1, b2)
    out = ctx.ransac_rel_pose(f1_t, f2_t, n_t, float(threshold), int(max_iterations), algorithm=REL_ALGORITHMS[name],
                              seed=_next_seed(), adaptive=True)
    ctx.synchronize()
    k = int(out["n_inliers"][0].item())
    return out["T"][0].cpu().numpy(), out["idx"][0, :k].cpu().numpy().astype(np.int64)


def triangulation_triangulate2(b1, b2, t12, R12):
    """-> X [n,3] in frame 1: OpenGV's closed-form midpoint of the two bearing rays."""
    ctx = _ctx()
    b1 = np.asarray(b1, dtype=np.float64)
    b2 = np.asarray(b2, dtype=np.float64)
    if b1.ndim != 2 or b1.shape[1] != 3 or b2.shape != b1.shape:
        raise ValueError("bearings must both be [n, 3]")
    if b1.shape[0] == 0:
        return np.empty((0, 3))
    X = ctx.triangulate2(_dev(ctx, b1, np.float64), _dev(ctx, b2, np.float64), np.asarray(t12, dtype=np.float64).reshape(3),
                         np.asarray(R12, dtype=np.float64).reshape(3, 3))
    ctx.synchronize()
    return X.cpu().numpy()
