"""Synthetic omnistereo frames for benchmarks and end-to-end tests (there is no network
for datasets).  A textured room is ray-cast through the GUM model of both mirrors: every omni pixel of
each mirror's annulus is lifted to its viewing ray (closed-form inverse of the distortion-free GUM),
the ray is intersected with the room's planes, and the hit point is coloured by a multi-scale
procedural checker texture hashed from its world coordinates.  Frames of one pair see the same room
from two poses, so stereo disparity and frame-to-frame motion are geometrically exact.  numpy only."""
import numpy as np


def lift_pixels(model, u, v):
    """Omni pixels -> unit viewing directions in the mirror frame.
    Inverse of GUM.get_pixel_from_3D_point_wrt_M: q = Ps - Cp = lambda * (x, y, sigma), |Ps| = 1; the radial polynomial
    rho_d = rho_u (1 + k1 rho_u^2 + k2 rho_u^4 + k3 rho_u^6) is inverted by Newton's method (to 1e-15 in a few steps for
    the mild distortions of a calibrated mirror)."""
    p = model.precalib_params
    y = (v - p.v_center) / p.gamma2
    x = (u - p.u_center - p.gamma1 * p.alpha_c * y) / p.gamma1
    if p.use_distortion and (p.k1 or p.k2 or p.k3):
        rd = np.sqrt(x * x + y * y)
        ru = rd.copy()
        for _ in range(30):
            r2 = ru * ru
            f = ru * (1.0 + r2 * (p.k1 + r2 * (p.k2 + r2 * p.k3))) - rd
            df = 1.0 + r2 * (3.0 * p.k1 + r2 * (5.0 * p.k2 + r2 * 7.0 * p.k3))
            ru = ru - f / df
        with np.errstate(divide="ignore", invalid="ignore"):
            sc = np.where(rd > 0, ru / rd, 1.0)
        x, y = x * sc, y * sc
    sigma = -1.0 if p.xi3 > 0 else 1.0  # sign of (Ps_z - xi3) for the visible half
    d = np.stack([x, y, np.full_like(x, sigma)], axis=-1)
    cp = np.array([p.xi1, p.xi2, p.xi3])
    a = np.sum(d * d, axis=-1)
    b = 2.0 * (d @ cp)
    c = cp @ cp - 1.0
    lam = (-b + np.sqrt(b * b - 4 * a * c)) / (2 * a)
    return cp + lam[..., None] * d


def _hash3(ix, iy, iz, salt):
    h = (ix.astype(np.int64) * 73856093) ^ (iy.astype(np.int64) * 19349663) ^ (iz.astype(np.int64) * 83492791) ^ salt
    h = (h ^ (h >> 13)) * 1274126177
    h = h ^ (h >> 16)
    return h & 0xFFFFFF


class Room(object):
    """Box room (mm), textured by hashed multi-scale cells; axis-aligned in its own frame, which is the world frame
    turned by yaw_deg about the vertical (so that a forward-looking pinhole camera at the identity pose sees a corner,
    i.e. non-coplanar structure, rather than a single wall)."""

    def __init__(self, seed=0, half_x=(2600.0, 3400.0), half_y=(2100.0, 3900.0), z_floor=-1400.0, z_ceil=1500.0,
                 cells=(260.0, 65.0), yaw_deg=0.0):
        rng = np.random.default_rng(seed)
        a = np.deg2rad(yaw_deg)
        self.R_room = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])  # room -> world
        self.seed = int(seed)
        self.planes = [(0, -rng.uniform(*half_x)), (0, rng.uniform(*half_x)), (1, -rng.uniform(*half_y)),
                       (1, rng.uniform(*half_y)), (2, z_floor), (2, z_ceil)]
        self.cells = cells

    def cast(self, origins, dirs):
        """origins [3], dirs [n,3] -> hit points [n,3] (nearest plane in front of the ray)."""
        t_best = np.full(dirs.shape[0], np.inf)
        o_room, d_room = np.asarray(origins) @ self.R_room, dirs @ self.R_room
        for axis, val in self.planes:
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (val - o_room[axis]) / d_room[:, axis]
            t = np.where(t > 1e-6, t, np.inf)
            t_best = np.minimum(t_best, t)
        return origins + dirs * t_best[:, None], t_best

    def colour(self, P):
        """World points [n,3] -> BGR uint8 [n,3]."""
        out = np.zeros((P.shape[0], 3), dtype=np.float64)
        P = P @ self.R_room
        weights = (0.62, 0.38)
        for lvl, (cell, w) in enumerate(zip(self.cells, weights)):
            idx = np.floor(P / cell + 0.5 * lvl).astype(np.int64)
            h = _hash3(idx[:, 0], idx[:, 1], idx[:, 2], 0x5bd1e995 * (self.seed + 1) + lvl)
            col = np.stack([(h >> s) & 0xFF for s in (0, 8, 16)], axis=-1).astype(np.float64)
            out += w * col
        return np.clip(out, 0, 255).astype(np.uint8)


def render_omni(gums, room, R, t, noise_sigma=2.0, rng=None):
    """One 8-bit BGR omni frame seen from the viewpoint X_world = R x + t (x in the rig frame [C], mm)."""
    W, H = gums.top_model.image_size
    img = np.zeros((H, W, 3), dtype=np.uint8)
    if gums.top_model.mask is None:
        gums.make_annulus_masks((H, W))
    for m in (gums.top_model, gums.bot_model):
        vv, uu = np.nonzero(m.mask)
        dirs = lift_pixels(m, uu.astype(np.float64), vv.astype(np.float64))  # mirror frame = [C] orientation
        origin = R @ m.F[:3, 0] + t
        P, _ = room.cast(origin, dirs @ R.T)
        img[vv, uu] = room.colour(P)
    if noise_sigma > 0:
        rng = rng or np.random.default_rng(0)
        img = np.clip(img.astype(np.float64) + rng.normal(0, noise_sigma, img.shape), 0, 255).astype(np.uint8)
        img[(gums.top_model.mask == 0) & (gums.bot_model.mask == 0)] = 0
    return img


def random_step(rng, max_t=100.0, max_deg=5.0):
    """Random SE(3) step (|t| <= max_t mm, angle <= max_deg): pose of the second viewpoint in the first."""
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(0.2, max_deg))
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = rng.normal(size=3)
    t = t / np.linalg.norm(t) * rng.uniform(0.1 * max_t, max_t)
    return R, t


def _render_pair(args):
    gums, seed, i, noise_sigma = args
    rng = np.random.default_rng(seed + i)
    room = Room(seed=seed + i)
    R, t = random_step(rng)
    ref = render_omni(gums, room, np.eye(3), np.zeros(3), noise_sigma, rng)
    cur = render_omni(gums, room, R, t, noise_sigma, rng)
    return ref, cur, (R, t)


_POOL_GUMS = None


def _render_pair_pooled(args):
    return _render_pair((_POOL_GUMS,) + args)


def make_frame_pairs(gums, n_pairs, seed=1234, noise_sigma=2.0, workers=1, first=0):
    """-> (omni [2*n_pairs, H, W, 3] u8, poses list of (R, t)): pair i = frames 2i (reference, identity pose)
    and 2i+1 (current, pose (R, t) in the reference frame); one room per pair.  workers > 1 renders the pairs in
    forked worker processes (same frames; call it BEFORE the process touches the GPU).  `first`: global index of
    pair 0 (a rank of a sharded job renders pairs first .. first + n_pairs - 1 of the job's sequence)."""
    global _POOL_GUMS
    W, H = gums.top_model.image_size
    if gums.top_model.mask is None:
        gums.make_annulus_masks((H, W))
    omni = np.zeros((2 * n_pairs, H, W, 3), dtype=np.uint8)
    poses = []
    jobs = [(seed, int(first) + i, noise_sigma) for i in range(n_pairs)]
    if workers > 1 and n_pairs > 1:
        import multiprocessing
        _POOL_GUMS = gums
        with multiprocessing.get_context("fork").Pool(min(int(workers), n_pairs)) as pool:
            results = pool.map(_render_pair_pooled, jobs, chunksize=max(1, n_pairs // (4 * int(workers))))
        _POOL_GUMS = None
    else:
        results = [_render_pair((gums,) + j) for j in jobs]
    for i, (ref, cur, pose) in enumerate(results):
        omni[2 * i], omni[2 * i + 1] = ref, cur
        poses.append(pose)
    return omni, poses


# ---- perspective RGB-D frames and on-disk sequences (for the demo CLIs / VO-loop tests) ---------------------
RGBD_AXES = np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]])  # camera (x right, y down, z forward) in the rig frame


def render_rgbd(room, R, t, rng, fx=554.256258, fy=554.256258, cx=319.5, cy=239.5, shape=(480, 640), depth_is_Z=True,
                holes=0.05, noise_sigma=2.0):
    """Pinhole view of the textured room from the rig pose X_world = R x + t (mm): BGR u8 [rows,cols,3] and depth
    f32 [rows,cols] in metres quantised to 1 mm (0 = no reading); radial distance when depth_is_Z is False."""
    v, u = np.mgrid[0:shape[0], 0:shape[1]]
    d = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones(shape)], axis=-1).reshape(-1, 3)
    P, tt = room.cast(np.asarray(t, dtype=np.float64), d @ (R @ RGBD_AXES).T)
    bgr = room.colour(P).reshape(shape + (3,))
    bgr = np.clip(bgr.astype(np.float64) + rng.normal(0, noise_sigma, bgr.shape), 0, 255).astype(np.uint8)
    Z = (tt * 1e-3).reshape(shape)  # the ray parameter with d_z = 1 is the Z depth
    depth = Z if depth_is_Z else Z * np.linalg.norm(d, axis=1).reshape(shape)
    depth = np.round(depth * 1000.0) / 1000.0
    depth[rng.random(shape) < holes] = 0.0
    return bgr, depth.astype(np.float32)


def trajectory(n_frames, seed=0, max_t=40.0, max_deg=2.0):
    """n rig poses (R, t [mm]) in the world: identity first, then a random walk of small SE(3) steps."""
    rng = np.random.default_rng(seed)
    R, t = np.eye(3), np.zeros(3)
    poses = [(R, t)]
    for _ in range(n_frames - 1):
        dR, dt = random_step(rng, max_t=max_t, max_deg=max_deg)
        R, t = R @ dR, R @ dt + t
        poses.append((R, t))
    return poses


def _render_seq_frame(args):
    gums, seed, R, t, k = args
    return render_omni(gums, Room(seed=seed), R, t, 2.0, np.random.default_rng(seed + 1 + k))


def _render_seq_frame_pooled(args):
    return _render_seq_frame((_POOL_GUMS,) + args)


def make_sequence(gums, n_frames, seed=0, max_t=40.0, max_deg=2.0, workers=1):
    """-> (omni [n_frames, H, W, 3] u8, poses list of (R, t [mm])): ONE room seen along a random-walk trajectory (frame k
    from pose k; pixel noise seeded per frame, so the frames do not depend on the number of workers).  workers > 1 renders
    in forked processes: call it BEFORE the process touches the GPU."""
    global _POOL_GUMS
    W, H = gums.top_model.image_size
    if gums.top_model.mask is None:
        gums.make_annulus_masks((H, W))
    poses = trajectory(n_frames, seed, max_t, max_deg)
    jobs = [(seed, R, t, k) for k, (R, t) in enumerate(poses)]
    if workers > 1 and n_frames > 1:
        import multiprocessing
        _POOL_GUMS = gums
        with multiprocessing.get_context("fork").Pool(min(int(workers), n_frames)) as pool:
            frames = pool.map(_render_seq_frame_pooled, jobs, chunksize=max(1, n_frames // (4 * int(workers))))
        _POOL_GUMS = None
    else:
        frames = [_render_seq_frame((gums,) + j) for j in jobs]
    return np.stack(frames), poses


def _render_rgbd_seq_frame(args):
    seed, R, t, k, depth_is_Z, yaw_deg = args
    room = Room(seed=seed, half_x=(1800.0, 2600.0), half_y=(2500.0, 3500.0), cells=(150.0, 40.0), yaw_deg=yaw_deg)
    return render_rgbd(room, R, t, np.random.default_rng(seed + 1 + k), depth_is_Z=depth_is_Z)


def make_rgbd_sequence(n_frames, seed=0, max_t=40.0, max_deg=2.0, depth_is_Z=False, yaw_deg=40.0, workers=1):
    """-> (bgr [n, 480, 640, 3] u8, depth [n, 480, 640] f32 metres, poses): one room along a random-walk trajectory, as
    write_rgbd_sequence renders it (frames do not depend on the number of workers; fork before the GPU is touched)."""
    poses = trajectory(n_frames, seed, max_t, max_deg)
    jobs = [(seed, R, t, k, depth_is_Z, yaw_deg) for k, (R, t) in enumerate(poses)]
    if workers > 1 and n_frames > 1:
        import multiprocessing
        with multiprocessing.get_context("fork").Pool(min(int(workers), n_frames)) as pool:
            frames = pool.map(_render_rgbd_seq_frame, jobs, chunksize=max(1, n_frames // (4 * int(workers))))
    else:
        frames = [_render_rgbd_seq_frame(j) for j in jobs]
    return np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), poses


def _write_gt_tum(filename, poses):
    from .omnistereo.transformations import quaternion_from_matrix
    with open(filename, "w") as f:
        f.write("# index tx ty tz qx qy qz qw   [m]\n")
        for i, (R, t) in enumerate(poses):
            T = np.eye(4)
            T[:3, :3] = R
            q = quaternion_from_matrix(T)
            f.write("%d %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n" % (i, t[0] * 1e-3, t[1] * 1e-3, t[2] * 1e-3, q[1], q[2], q[3], q[0]))


def write_sos_sequence(path, gums, n_frames=6, seed=0, max_t=40.0, max_deg=2.0):
    """<path>/omni/image-%04d.png + gt_TUM.txt, <path>/gums-calibrated.json: what demo_vo_sos.py reads."""
    import os
    from .omnistereo.common_cv import imwrite
    from .omnistereo.gum import save_gums_json
    os.makedirs(os.path.join(path, "omni"), exist_ok=True)
    room = Room(seed=seed)
    poses = trajectory(n_frames, seed, max_t, max_deg)
    rng = np.random.default_rng(seed + 1)
    for i, (R, t) in enumerate(poses):
        imwrite(os.path.join(path, "omni", "image-%04d.png" % i), render_omni(gums, room, R, t, 2.0, rng))
    _write_gt_tum(os.path.join(path, "omni", "gt_TUM.txt"), poses)
    save_gums_json(gums, os.path.join(path, "gums-calibrated.json"))
    return poses


def write_rgbd_sequence(path, n_frames=6, seed=0, max_t=40.0, max_deg=2.0, depth_is_Z=False, yaw_deg=40.0):
    """<path>/rgbd/rgb/%04d.png, <path>/rgbd/depth/%04d.png (16-bit, mm) + gt_TUM.txt: what demo_vo_rgbd.py reads
    (its synthetic setting stores RADIAL depth, demo_vo_rgbd.py:68)."""
    import os
    from PIL import Image
    from .omnistereo.common_cv import imwrite
    for sub in ("rgb", "depth"):
        os.makedirs(os.path.join(path, "rgbd", sub), exist_ok=True)
    room = Room(seed=seed, half_x=(1800.0, 2600.0), half_y=(2500.0, 3500.0), cells=(150.0, 40.0), yaw_deg=yaw_deg)
    poses = trajectory(n_frames, seed, max_t, max_deg)
    rng = np.random.default_rng(seed + 1)
    for i, (R, t) in enumerate(poses):
        bgr, depth = render_rgbd(room, R, t, rng, depth_is_Z=depth_is_Z)
        imwrite(os.path.join(path, "rgbd", "rgb", "%04d.png" % i), bgr)
        Image.fromarray(np.round(depth * 1000.0).astype(np.uint16)).save(os.path.join(path, "rgbd", "depth", "%04d.png" % i))
    _write_gt_tum(os.path.join(path, "rgbd", "gt_TUM.txt"), poses)
    return poses
