"""ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy/ctypes face of the CPU restatement in oracle/*.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only
as the checker; nothing under vo_single_camera_sos_amd/ imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "build", "libsosvo_oracle.so")
# SOSVO_ORACLE_LIB: another build of the same sources (the sanitizer build of `make -C oracle sanitize`,
# scripts/oracle_sanitize.sh); never set by the package, the bench or the GPU tests.
_LIB_OVERRIDE = os.environ.get("SOSVO_ORACLE_LIB")
_lib = None

KEY_SHIFT = 20
KEY_IDX_MASK = 0xFFFFF
KEY_NONE = 0xFFFFFFFF


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_OVERRIDE or LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# ---- K7 ---------------------------------------------------------------------------------
def match_hamming(q, t, k=1):
    """q [nq,32] u8, t [nt,32] u8 -> keys [nq,k] u32 = (dist << 20 | train idx)."""
    q = _c(q, np.uint8).reshape(-1, 32)
    t = _c(t, np.uint8).reshape(-1, 32)
    keys = np.empty((q.shape[0], k), dtype=np.uint32)
    lib().orc_match_hamming(_p(q), _p(t), ctypes.c_int32(q.shape[0]), ctypes.c_int32(t.shape[0]),
                            ctypes.c_int32(k), _p(keys))
    return keys


def match_l2(q, t, k=1):
    """q [nq,dim] f32, t [nt,dim] f32 -> keys [nq,k] u64 = (float32 distance bits << 32 | train idx)."""
    q = _c(q, np.float32)
    t = _c(t, np.float32)
    keys = np.empty((q.shape[0], k), dtype=np.uint64)
    lib().orc_match_l2(_p(q), _p(t), ctypes.c_int32(q.shape[0]), ctypes.c_int32(t.shape[0]), ctypes.c_int32(q.shape[1]),
                       ctypes.c_int32(k), _p(keys))
    return keys


def match_radius(q, t, max_distance, cap):
    """q [nq,32] u8, t [nt,32] u8 -> (keys [nq,cap] u32 ascending, KEY_NONE padded; counts [nq] i32)."""
    q = _c(q, np.uint8).reshape(-1, 32)
    t = _c(t, np.uint8).reshape(-1, 32)
    keys = np.empty((q.shape[0], cap), dtype=np.uint32)
    counts = np.empty((q.shape[0],), dtype=np.int32)
    lib().orc_match_radius(_p(q), _p(t), ctypes.c_int32(q.shape[0]), ctypes.c_int32(t.shape[0]),
                           ctypes.c_int32(int(max_distance)), ctypes.c_int32(cap), _p(keys), _p(counts))
    return keys, counts


def sort_matches(keys):
    """keys [nq] or [nq,1] u32 -> order [nq] i32, stable by distance."""
    keys = _c(keys, np.uint32).reshape(-1)
    order = np.empty(keys.shape[0], dtype=np.int32)
    lib().orc_sort_matches(_p(keys), ctypes.c_int32(keys.shape[0]), _p(order))
    return order


# ---- K8/K9/K10: absolute-pose RANSAC and refinement ----------------------------------------
def _opt_cam(cam, cam_off, cam_rot):
    if cam is None:
        return None, None, None, 1
    cam = _c(cam, np.int32).reshape(-1)
    cam_off = _c(cam_off, np.float64).reshape(-1, 3)
    cam_rot = _c(cam_rot, np.float64).reshape(-1, 3, 3)
    return cam, cam_off, cam_rot, cam_off.shape[0]


def _pn(a):
    return _p(a) if a is not None else ctypes.c_void_p(0)


def score_points(f, p, T, cam=None, cam_off=None, cam_rot=None):
    f = _c(f, np.float64).reshape(-1, 3)
    p = _c(p, np.float64).reshape(-1, 3)
    T = _c(T, np.float64).reshape(3, 4)
    cam, cam_off, cam_rot, _ = _opt_cam(cam, cam_off, cam_rot)
    out = np.empty(f.shape[0], dtype=np.float64)
    lib().orc_score_points(_p(f), _p(p), _pn(cam), _pn(cam_off), _pn(cam_rot), ctypes.c_int32(f.shape[0]),
                           _p(T), _p(out))
    return out


def ransac_abs_pose(f, p, thr, max_iter, seed=0, adaptive=False, cam=None, cam_off=None, cam_rot=None,
                    want_counts=False, epnp=False, gp3p=False, twopt=False):
    """-> dict(T [3,4], mask [n] bool, n_inliers, best_iter, iters_used, status[, counts]).  epnp: central problems
    only, hypotheses from EPnP on 6-point samples instead of Kneip P3P + a 4th point."""
    f = _c(f, np.float64).reshape(-1, 3)
    p = _c(p, np.float64).reshape(-1, 3)
    n = f.shape[0]
    cam, cam_off, cam_rot, ncam = _opt_cam(cam, cam_off, cam_rot)
    T = np.zeros((3, 4), dtype=np.float64)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    n_inl = ctypes.c_int32(0)
    best_it = ctypes.c_int32(0)
    used = ctypes.c_int32(0)
    counts = np.zeros(max(max_iter, 1), dtype=np.int32) if want_counts else None
    L = lib()
    L.orc_ransac_abs_pose.restype = ctypes.c_int32
    st = L.orc_ransac_abs_pose(_p(f), _p(p), _pn(cam), _pn(cam_off), _pn(cam_rot), ctypes.c_int32(n),
                               ctypes.c_int32(ncam), ctypes.c_double(thr), ctypes.c_int32(max_iter),
                               ctypes.c_int32((1 if adaptive else 0) | (2 if epnp else 0) | (4 if gp3p else 0) | (8 if twopt else 0)), ctypes.c_uint64(seed), _p(T), _p(mask),
                               ctypes.byref(n_inl), ctypes.byref(best_it), ctypes.byref(used), _pn(counts))
    out = dict(T=T, mask=mask[:n].astype(bool), n_inliers=n_inl.value, best_iter=best_it.value,
               iters_used=used.value, status=st)
    if want_counts:
        out["counts"] = counts[:max_iter]
    return out


def epnp(f, p):
    """EPnP on 5..8 correspondences -> T [3,4] (pose of the camera in the world) or None."""
    f = _c(f, np.float64).reshape(-1, 3)
    p = _c(p, np.float64).reshape(-1, 3)
    T = np.zeros((3, 4), dtype=np.float64)
    L = lib()
    L.orc_epnp_solve.restype = ctypes.c_int32
    ok = L.orc_epnp_solve(_p(f), _p(p), ctypes.c_int32(f.shape[0]), _p(T))
    return T if ok else None


def symeig12(A):
    """Eigen-decomposition of a symmetric 12 x 12 matrix by the EPnP restatement's solver (Householder + implicit QL)
    -> (eigenvalues [12] unordered, eigenvectors as columns [12,12]) or None."""
    V = np.array(A, dtype=np.float64, order="C").reshape(12, 12).copy()
    d = np.zeros(12, dtype=np.float64)
    L = lib()
    L.orc_symeig12_solve.restype = ctypes.c_int32
    ok = L.orc_symeig12_solve(_p(V), _p(d))
    return (d, V) if ok else None


def ransac_rel_pose(f1, f2, thr, max_iter, seed=0, adaptive=False, want_counts=False, algorithm=8):
    """2D-2D relative-pose RANSAC with the eight-point solver -> dict(T [3,4] (pose of view 2 in frame 1, |t| = 1), mask,
    n_inliers, best_iter, iters_used, status[, counts])."""
    f1 = _c(f1, np.float64).reshape(-1, 3)
    f2 = _c(f2, np.float64).reshape(-1, 3)
    n = f1.shape[0]
    T = np.zeros((3, 4), dtype=np.float64)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    n_inl, best_it, used = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
    counts = np.zeros(max(max_iter, 1), dtype=np.int32) if want_counts else None
    L = lib()
    L.orc_ransac_rel_pose.restype = ctypes.c_int32
    st = L.orc_ransac_rel_pose(_p(f1), _p(f2), ctypes.c_int32(n), ctypes.c_int32(algorithm), ctypes.c_double(thr), ctypes.c_int32(max_iter),
                               ctypes.c_int32(1 if adaptive else 0), ctypes.c_uint64(seed), _p(T), _p(mask), ctypes.byref(n_inl),
                               ctypes.byref(best_it), ctypes.byref(used), _pn(counts))
    out = dict(T=T, mask=mask[:n].astype(bool), n_inliers=n_inl.value, best_iter=best_it.value, iters_used=used.value, status=st)
    if want_counts:
        out["counts"] = counts[:max_iter]
    return out


def rel_score(T, f1, f2):
    """The reference's relative-pose score (pose_est_tools.py:150-203) of one correspondence."""
    L = lib()
    L.orc_rel_score_once.restype = ctypes.c_double
    return L.orc_rel_score_once(_p(_c(T, np.float64)), _p(_c(f1, np.float64)), _p(_c(f2, np.float64)))


def eightpt(f1, f2):
    T = np.zeros((3, 4), dtype=np.float64)
    L = lib()
    L.orc_eightpt_solve.restype = ctypes.c_int32
    ok = L.orc_eightpt_solve(_p(_c(f1, np.float64).reshape(8, 3)), _p(_c(f2, np.float64).reshape(8, 3)), _p(T))
    return T if ok else None


def fivept(f1, f2):
    """Nister's five-point algorithm: f1, f2 [5,3] -> E [k,3,3], k <= 10 real solutions (f1^T E f2 = 0)."""
    E = np.zeros((10, 9), dtype=np.float64)
    L = lib()
    L.orc_fivept_solve.restype = ctypes.c_int32
    k = L.orc_fivept_solve(_p(_c(f1, np.float64).reshape(5, 3)), _p(_c(f2, np.float64).reshape(5, 3)), _p(E))
    return E[:k].reshape(k, 3, 3)


def sevenpt(f1, f2):
    """Seven-point algorithm: f1, f2 [7,3] -> E [k,3,3], k in (1, 3)."""
    E = np.zeros((3, 9), dtype=np.float64)
    L = lib()
    L.orc_sevenpt_solve.restype = ctypes.c_int32
    k = L.orc_sevenpt_solve(_p(_c(f1, np.float64).reshape(7, 3)), _p(_c(f2, np.float64).reshape(7, 3)), _p(E))
    return E[:k].reshape(k, 3, 3)


def gp3p(fb, o, P, want_octic=False):
    """Generalised P3P: fb, o, P [3,3] (body-frame unit bearings, their camera offsets, world points) -> list of T [3,4]
    (P = R x_body + t), optionally with the octic's coefficients (lowest first, lengths in units of the largest side)."""
    fb, o, P = (_c(a, np.float64).reshape(3, 3) for a in (fb, o, P))
    T = np.zeros((8, 3, 4), dtype=np.float64)
    oct_ = np.zeros(9, dtype=np.float64)
    L = lib()
    L.orc_gp3p_solve.restype = ctypes.c_int32
    ns = L.orc_gp3p_solve(_p(fb), _p(o), _p(P), _p(T), _p(oct_))
    sols = [T[k].copy() for k in range(ns)]
    return (sols, oct_) if want_octic else sols


def jacobi12(A):
    """The eigen-solver EPnP uses (round-robin Jacobi) -> (eigenvalues [12] unordered, eigenvectors as columns [12,12])."""
    W = np.array(A, dtype=np.float64, order="C").reshape(12, 12).copy()
    d, V = np.zeros(12, dtype=np.float64), np.zeros((12, 12), dtype=np.float64)
    L = lib()
    L.orc_jacobi12_solve.restype = ctypes.c_int32
    L.orc_jacobi12_solve(_p(W), _p(d), _p(V))
    return d, V


def sample_distinct(n, k, seed, it):
    s = np.zeros(k, dtype=np.int32)
    L = lib()
    L.orc_sample_distinct_once.restype = ctypes.c_int32
    ok = L.orc_sample_distinct_once(ctypes.c_int32(n), ctypes.c_int32(k), ctypes.c_uint64(seed), ctypes.c_int32(it), _p(s))
    return s if ok else None


def hypothesis_once(f, p, seed, it, cam=None, cam_off=None, cam_rot=None):
    f = _c(f, np.float64).reshape(-1, 3)
    p = _c(p, np.float64).reshape(-1, 3)
    cam, cam_off, cam_rot, ncam = _opt_cam(cam, cam_off, cam_rot)
    T = np.zeros((3, 4), dtype=np.float64)
    s4 = np.zeros(4, dtype=np.int32)
    L = lib()
    L.orc_hypothesis_once.restype = ctypes.c_int32
    ok = L.orc_hypothesis_once(_p(f), _p(p), _pn(cam), _pn(cam_off), _pn(cam_rot), ctypes.c_int32(f.shape[0]),
                               ctypes.c_int32(ncam), ctypes.c_uint64(seed), ctypes.c_int32(it), _p(T), _p(s4))
    return bool(ok), T, s4


def p3p_kneip(f3, p3):
    f3 = _c(f3, np.float64).reshape(3, 3)
    p3 = _c(p3, np.float64).reshape(3, 3)
    R = np.zeros((4, 3, 3))
    C = np.zeros((4, 3))
    L = lib()
    L.orc_p3p_kneip.restype = ctypes.c_int32
    n = L.orc_p3p_kneip(_p(f3), _p(p3), _p(R), _p(C))
    return R[:n], C[:n]


def quartic_real_roots(a5):
    a5 = _c(a5, np.float64).reshape(5)
    r = np.zeros(4)
    L = lib()
    L.orc_quartic_real_roots.restype = ctypes.c_int32
    n = L.orc_quartic_real_roots(_p(a5), _p(r))
    return r[:n]


def refine_abs_pose(f, p, T0, idx=None, cam=None, cam_off=None, cam_rot=None, max_lm_iter=30):
    f = _c(f, np.float64).reshape(-1, 3)
    p = _c(p, np.float64).reshape(-1, 3)
    cam, cam_off, cam_rot, _ = _opt_cam(cam, cam_off, cam_rot)
    T = _c(T0, np.float64).reshape(3, 4).copy()
    if idx is not None:
        idx = _c(idx, np.int32).reshape(-1)
    cost = ctypes.c_double(0)
    iters = ctypes.c_int32(0)
    lib().orc_refine_abs_pose(_p(f), _p(p), _pn(cam), _pn(cam_off), _pn(cam_rot), ctypes.c_int32(f.shape[0]),
                              _pn(idx), ctypes.c_int32(0 if idx is None else idx.shape[0]), _p(T),
                              ctypes.c_int32(max_lm_iter), ctypes.byref(cost), ctypes.byref(iters))
    return T, cost.value, iters.value


# ---- geometry (the reference's own numpy code, pinned by tests/golden) ------------------------
def pano_to_angles(u, v, cols, rows, pixel_size, h_max):
    u = _c(u, np.float64).reshape(-1)
    v = _c(v, np.float64).reshape(-1)
    pano = np.array([cols, rows, pixel_size, h_max], dtype=np.float64)
    az = np.empty_like(u)
    el = np.empty_like(u)
    lib().orc_pano_to_angles(_p(u), _p(v), ctypes.c_int32(u.shape[0]), _p(pano), _p(az), _p(el))
    return az, el


def angles_to_bearing(az, el):
    az = _c(az, np.float64).reshape(-1)
    el = _c(el, np.float64).reshape(-1)
    b = np.empty((az.shape[0], 3), dtype=np.float64)
    lib().orc_angles_to_bearing(_p(az), _p(el), ctypes.c_int32(az.shape[0]), _p(b))
    return b


def triangulate_midpoint(az1, el1, az2, el2, F1, F2):
    az1, el1, az2, el2 = [_c(a, np.float64).reshape(-1) for a in (az1, el1, az2, el2)]
    F1 = _c(F1, np.float64).reshape(3)
    F2 = _c(F2, np.float64).reshape(3)
    X = np.empty((az1.shape[0], 3), dtype=np.float64)
    lib().orc_triangulate_midpoint(_p(az1), _p(el1), _p(az2), _p(el2), ctypes.c_int32(az1.shape[0]), _p(F1), _p(F2),
                                   _p(X))
    return X


def triangulate2(b1, b2, t12, R12):
    b1 = _c(b1, np.float64).reshape(-1, 3)
    b2 = _c(b2, np.float64).reshape(-1, 3)
    t12 = _c(t12, np.float64).reshape(3)
    R12 = _c(R12, np.float64).reshape(9)
    X = np.empty_like(b1)
    lib().orc_triangulate2(_p(b1), _p(b2), ctypes.c_int32(b1.shape[0]), _p(t12), _p(R12), _p(X))
    return X


def range_filter_homo(X, min_range, max_range):
    X = _c(X, np.float64).reshape(-1, 3)
    ok = np.empty(X.shape[0], dtype=np.uint8)
    lib().orc_range_filter_homo(_p(X), ctypes.c_int32(X.shape[0]), ctypes.c_double(min_range),
                                ctypes.c_double(max_range), _p(ok))
    return ok.astype(bool)


def pixel_gates(top_uv, bot_uv, min_disp, max_hdiff):
    top_uv = _c(top_uv, np.float64).reshape(-1, 2)
    bot_uv = _c(bot_uv, np.float64).reshape(-1, 2)
    ok = np.empty(top_uv.shape[0], dtype=np.uint8)
    lib().orc_pixel_gates(_p(top_uv), _p(bot_uv), ctypes.c_int32(top_uv.shape[0]), ctypes.c_double(min_disp),
                          ctypes.c_double(max_hdiff), _p(ok))
    return ok.astype(bool)


def rgbd_backproject(depth, u, v, intr, depth_is_Z):
    depth = _c(depth, np.float32)
    u = _c(u, np.int32).reshape(-1)
    v = _c(v, np.int32).reshape(-1)
    intr = _c(intr, np.float64).reshape(5)
    xyz = np.empty((u.shape[0], 3), dtype=np.float64)
    b = np.empty((u.shape[0], 3), dtype=np.float64)
    lib().orc_rgbd_backproject(_p(depth), ctypes.c_int32(depth.shape[0]), ctypes.c_int32(depth.shape[1]), _p(u), _p(v),
                               ctypes.c_int32(u.shape[0]), _p(intr), ctypes.c_int32(1 if depth_is_Z else 0), _p(xyz),
                               _p(b))
    return xyz, b


def unwrap_lut(gum13, cols, rows, pixel_size, h_max, h_min, elev_low, elev_high):
    gum13 = _c(gum13, np.float64).reshape(13)
    pano = np.array([cols, rows, pixel_size, h_max, h_min], dtype=np.float64)
    elev = np.array([elev_low, elev_high], dtype=np.float64)
    mx = np.empty((int(rows), int(cols)), dtype=np.float64)
    my = np.empty((int(rows), int(cols)), dtype=np.float64)
    lib().orc_unwrap_lut(_p(gum13), _p(pano), _p(elev), _p(mx), _p(my))
    return mx, my


# ---- image stages (OpenCV semantics restated, see oracle/image.c) ---------------------------------
def unwrap(omni, mask, map_x, map_y):
    """omni [H,W,3] u8, mask [H,W] u8 or None, map_x/map_y [rows,cols] f32 -> pano [rows,cols,3] u8."""
    omni = _c(omni, np.uint8)
    H, W = omni.shape[:2]
    map_x = _c(map_x, np.float32)
    map_y = _c(map_y, np.float32)
    rows, cols = map_x.shape
    if mask is not None:
        mask = _c(mask, np.uint8)
    pano = np.empty((rows, cols, 3), dtype=np.uint8)
    lib().orc_unwrap(_p(omni), _pn(mask), ctypes.c_int32(H), ctypes.c_int32(W), _p(map_x), _p(map_y),
                     ctypes.c_int32(rows), ctypes.c_int32(cols), _p(pano))
    return pano


def median_gray(pano, ksize, want_bgr=False):
    pano = _c(pano, np.uint8)
    rows, cols = pano.shape[:2]
    gray = np.empty((rows, cols), dtype=np.uint8)
    bgr = np.empty((rows, cols, 3), dtype=np.uint8) if want_bgr else None
    lib().orc_median_gray(_p(pano), ctypes.c_int32(rows), ctypes.c_int32(cols), ctypes.c_int32(ksize), _p(gray),
                          _pn(bgr))
    return (gray, bgr) if want_bgr else gray


def min_eigen(gray):
    gray = _c(gray, np.uint8)
    eig = np.empty(gray.shape, dtype=np.float32)
    lib().orc_min_eigen(_p(gray), ctypes.c_int32(gray.shape[0]), ctypes.c_int32(gray.shape[1]), _p(eig))
    return eig


def gft_select(eig, mask_bits, which, quality=0.01, min_distance=5.0, max_corners=1000):
    eig = _c(eig, np.float32)
    mask_bits = _c(mask_bits, np.uint32)
    rows, cols = eig.shape
    kp = np.zeros((rows * cols, 2), dtype=np.float32)
    maxv = ctypes.c_float(0)
    L = lib()
    L.orc_gft_select.restype = ctypes.c_int32
    n = L.orc_gft_select(_p(eig), _p(mask_bits), ctypes.c_int32(which), ctypes.c_int32(rows), ctypes.c_int32(cols),
                         ctypes.c_double(quality), ctypes.c_double(min_distance), ctypes.c_int32(max_corners), _p(kp),
                         ctypes.byref(maxv))
    return kp[:n].copy(), maxv.value


def gauss7(gray):
    gray = _c(gray, np.uint8)
    out = np.empty_like(gray)
    lib().orc_gauss7(_p(gray), ctypes.c_int32(gray.shape[0]), ctypes.c_int32(gray.shape[1]), _p(out))
    return out


def orb_pattern():
    """The default descriptor pattern: OpenCV's bit_pattern_31_ ([512, 2] int8, test t compares points 2t and 2t + 1)."""
    pts = np.empty((512, 2), dtype=np.int8)
    lib().orc_orb_pattern_opencv(_p(pts))
    return pts


def orb_pattern_seeded():
    """The seeded pattern of rounds 1-3 (|coordinate| <= 12)."""
    pts = np.empty((512, 2), dtype=np.int8)
    lib().orc_orb_pattern(_p(pts))
    return pts


def orb_level_size(rows, cols, level):
    r, c = ctypes.c_int32(0), ctypes.c_int32(0)
    lib().orc_orb_level_size(ctypes.c_int32(rows), ctypes.c_int32(cols), ctypes.c_int32(level), ctypes.byref(r),
                             ctypes.byref(c))
    return r.value, c.value


def resize_linear(src, h1, w1):
    src = _c(src, np.uint8)
    dst = np.empty((h1, w1), dtype=np.uint8)
    lib().orc_resize_linear(_p(src), ctypes.c_int32(src.shape[0]), ctypes.c_int32(src.shape[1]), _p(dst),
                            ctypes.c_int32(h1), ctypes.c_int32(w1))
    return dst


def fast_score_map(img, thr=20):
    img = _c(img, np.uint8)
    out = np.empty_like(img)
    lib().orc_fast_score_map(_p(img), ctypes.c_int32(img.shape[0]), ctypes.c_int32(img.shape[1]), ctypes.c_int32(thr),
                             _p(out))
    return out


def agast_nms(score):
    """AGAST's block-maximum non-maximum suppression on a corner score map -> bool [rows, cols]."""
    score = _c(score, np.uint8)
    keep = np.zeros(score.shape, dtype=np.uint8)
    lib().orc_agast_nms(_p(score), ctypes.c_int32(score.shape[0]), ctypes.c_int32(score.shape[1]), _p(keep))
    return keep.astype(bool)


def orb_quotas(nfeatures):
    q = np.zeros(8, dtype=np.int32)
    lib().orc_orb_quotas(ctypes.c_int32(nfeatures), _p(q))
    return q


def orb_detect(gray, mask_bits, nmask, nfeatures, cap=1024):
    """-> list over masks of (kp [n,4] = x, y, angle_deg, level; resp [n])."""
    gray = _c(gray, np.uint8)
    mask_bits = _c(mask_bits, np.uint32)
    rows, cols = gray.shape
    kp = np.zeros((nmask, cap, 4), dtype=np.float32)
    resp = np.zeros((nmask, cap), dtype=np.float32)
    n = np.zeros(nmask, dtype=np.int32)
    lib().orc_orb_detect(_p(gray), ctypes.c_int32(rows), ctypes.c_int32(cols), _p(mask_bits), ctypes.c_int32(nmask),
                         ctypes.c_int32(nfeatures), ctypes.c_int32(cap), _p(kp), _p(resp), _p(n))
    return [(kp[m, : n[m]].copy(), resp[m, : n[m]].copy()) for m in range(nmask)]


def orb_describe_levels(gray, kp4, pattern=None):
    """kp4 [n,4] = (x, y, angle_deg, level) -> (desc [m,32], kept_idx [m])."""
    gray = _c(gray, np.uint8)
    kp4 = _c(kp4, np.float32).reshape(-1, 4)
    if pattern is None:
        pattern = orb_pattern()
    pattern = _c(pattern, np.int8)
    n = kp4.shape[0]
    desc = np.zeros((max(n, 1), 32), dtype=np.uint8)
    kept = np.zeros(max(n, 1), dtype=np.int32)
    L = lib()
    L.orc_orb_describe_levels.restype = ctypes.c_int32
    m = L.orc_orb_describe_levels(_p(gray), ctypes.c_int32(gray.shape[0]), ctypes.c_int32(gray.shape[1]), _p(kp4),
                                  ctypes.c_int32(n), _p(pattern), _p(desc), _p(kept))
    return desc[:m].copy(), kept[:m].copy()


def orb_describe(blurred, kp_xy, cos_a, sin_a, pattern=None, edge=31):
    """-> (desc [m,32] u8, kept_idx [m] i32): keypoints within `edge` px of the border are dropped."""
    blurred = _c(blurred, np.uint8)
    kp_xy = _c(kp_xy, np.float32).reshape(-1, 2)
    if pattern is None:
        pattern = orb_pattern()
    pattern = _c(pattern, np.int8)
    n = kp_xy.shape[0]
    desc = np.zeros((max(n, 1), 32), dtype=np.uint8)
    kept = np.zeros(max(n, 1), dtype=np.int32)
    L = lib()
    L.orc_orb_describe.restype = ctypes.c_int32
    m = L.orc_orb_describe(_p(blurred), ctypes.c_int32(blurred.shape[0]), ctypes.c_int32(blurred.shape[1]), _p(kp_xy),
                           ctypes.c_int32(n), ctypes.c_float(cos_a), ctypes.c_float(sin_a), _p(pattern),
                           ctypes.c_int32(edge), _p(desc), _p(kept))
    return desc[:m].copy(), kept[:m].copy()
