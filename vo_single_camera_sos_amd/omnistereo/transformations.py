"""Host-side mirror of the few helpers of omnistereo/transformations.py that the VO loop uses
(pose_est_tools.py:1264-1678): homogeneous-matrix composition and inversion, matrix <-> quaternion
([w, x, y, z] order, as in that module), TUM entries, and the relative-pose-error metrics that drive the
keyframe policy.  Own numpy implementations of the published formulas; pinned by tests/golden/transforms.npz
(values captured from the reference module) and the reference's doctest values."""
import numpy as np

_EPS = np.finfo(float).eps * 4.0


def identity_matrix():
    """transformations.py:209"""
    return np.identity(4)


def translation_matrix(direction):
    """transformations.py:222"""
    M = np.identity(4)
    M[:3, 3] = np.asarray(direction, dtype=np.float64)[:3]
    return M


def translation_from_matrix(matrix):
    """transformations.py:236"""
    return np.array(matrix, copy=False)[:3, 3].copy()


def concatenate_matrices(*matrices):
    """transformations.py:1803: the product M0 . M1 . ... of homogeneous matrices."""
    M = np.identity(4)
    for i in matrices:
        M = np.dot(M, i)
    return M


def inverse_matrix(matrix):
    """transformations.py:1786 (general inverse, as there)."""
    return np.linalg.inv(matrix)


def rotation_matrix(angle, direction, point=None):
    """transformations.py:307: rotation about the axis `direction` through `point` (Rodrigues)."""
    sina, cosa = np.sin(angle), np.cos(angle)
    d = np.asarray(direction, dtype=np.float64)[:3]
    d = d / np.linalg.norm(d)
    R = np.diag([cosa, cosa, cosa])
    R += np.outer(d, d) * (1.0 - cosa)
    d = d * sina
    R += np.array([[0.0, -d[2], d[1]], [d[2], 0.0, -d[0]], [-d[1], d[0], 0.0]])
    M = np.identity(4)
    M[:3, :3] = R
    if point is not None:
        point = np.asarray(point[:3], dtype=np.float64)
        M[:3, 3] = point - np.dot(R, point)
    return M


def _rotation_from_scaled_quaternion(w, x, y, z):
    """3x3 rotation from a quaternion scaled to norm sqrt(2) (every product then carries the factor 2)."""
    return np.array([[1.0 - y * y - z * z, x * y - z * w, x * z + y * w],
                     [x * y + z * w, 1.0 - x * x - z * z, y * z - x * w],
                     [x * z - y * w, y * z + x * w, 1.0 - x * x - y * y]])


def quaternion_matrix(quaternion):
    """transformations.py:1214: [w, x, y, z] -> 4x4 rotation matrix (identity for a null quaternion)."""
    q = np.array(quaternion, dtype=np.float64, copy=True)
    n = np.dot(q, q)
    T = np.identity(4)
    if n >= _EPS:
        w, x, y, z = q * np.sqrt(2.0 / n)
        T[:3, :3] = _rotation_from_scaled_quaternion(w, x, y, z)
    return T


def quaternion_from_matrix(matrix, isprecise=False):
    """transformations.py:1258: rotation part of `matrix` -> unit quaternion [w, x, y, z], w >= 0, as the
    eigenvector for the largest eigenvalue of Bar-Itzhack's symmetric 4x4 matrix
        K = 1/3 [[R + R^T - tr(R) I, z], [z^T, tr(R)]],  z = (R21 - R12, R02 - R20, R10 - R01)
    (robust to slightly non-orthonormal input; this is the isprecise=False path, the only one the VO loop
    uses, pose_est_tools.py:1611).  isprecise is accepted for signature compatibility and ignored: for an exactly
    orthonormal matrix both of the reference's paths give this quaternion."""
    R = np.array(matrix, dtype=np.float64, copy=False)[:3, :3]
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    K = np.empty((4, 4))
    K[:3, :3] = R + R.T
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        K[i, i] = R[i, i] - R[j, j] - R[k, k]
    K[3, :3] = K[:3, 3] = (R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1])
    K[3, 3] = tr
    evals, evecs = np.linalg.eigh(K / 3.0)          # LAPACK reads the lower triangle
    x, y, z, w = evecs[:, np.argmax(evals)]
    q = np.array([w, x, y, z])
    return -q if q[0] < 0.0 else q


def quaternions_from_matrices(matrices):
    """quaternion_from_matrix for a stack [n, 4, 4] (or [n, 3, 3]) -> [n, 4] rows [w, x, y, z]: the same float64 operations
    element by element and the same LAPACK call per matrix (numpy's eigh loops over the stack), so every row equals the
    one-at-a-time result bit for bit (tests/test_host_vo_helpers.py) -- run_VO formats a window's pose lines with it."""
    M = np.asarray(matrices, dtype=np.float64)
    n = M.shape[0]
    if n == 0:
        return np.zeros((0, 4))
    R = M[:, :3, :3]
    K = np.empty((n, 4, 4))
    K[:, :3, :3] = R + R.transpose(0, 2, 1)
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        K[:, i, i] = R[:, i, i] - R[:, j, j] - R[:, k, k]
    z = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
    K[:, 3, :3] = z
    K[:, :3, 3] = z
    K[:, 3, 3] = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    evals, evecs = np.linalg.eigh(K / 3.0)
    v = evecs[np.arange(n), :, np.argmax(evals, axis=1)]        # [n, 4] = (x, y, z, w)
    q = v[:, [3, 0, 1, 2]]
    return np.where(q[:, :1] < 0.0, -q, q)


def quaternion_multiply(q1, q0):
    """Hamilton product q1 * q0 of two [w, x, y, z] quaternions (the rotation q0 followed by q1)."""
    w0, x0, y0, z0 = q0
    w1, x1, y1, z1 = q1
    return np.array([w1 * w0 - x1 * x0 - y1 * y0 - z1 * z0, w1 * x0 + x1 * w0 + y1 * z0 - z1 * y0,
                     w1 * y0 - x1 * z0 + y1 * w0 + z1 * x0, w1 * z0 + x1 * y0 - y1 * x0 + z1 * w0], dtype=np.float64)


def quaternion_from_euler(ai, aj, ak, axes="sxyz"):
    """transformations.py:1146 for the one convention this path uses ("sxyz": rotations about the STATIC x, then y, then z
    axes, the POV-Ray pose files of common_tools.get_poses_from_file): the product q_z(ak) q_y(aj) q_x(ai),
    [w, x, y, z], not sign-normalised."""
    if axes != "sxyz":
        raise NotImplementedError("quaternion_from_euler: only the static x-y-z convention ('sxyz') is built")
    qx = np.array([np.cos(ai / 2.0), np.sin(ai / 2.0), 0.0, 0.0])
    qy = np.array([np.cos(aj / 2.0), 0.0, np.sin(aj / 2.0), 0.0])
    qz = np.array([np.cos(ak / 2.0), 0.0, 0.0, np.sin(ak / 2.0)])
    return quaternion_multiply(qz, quaternion_multiply(qy, qx))


def pose_matrix_from_quaternion_and_translation(q, t):
    T = quaternion_matrix(q)
    T[:3, 3] = np.asarray(t, dtype=np.float64)[:3]
    return T


def transform44_from_TUM_entry(l, scale_translation=1.0, has_timestamp=True):
    """transformations.py:2186: ([stamp,] tx, ty, tz, qx, qy, qz, qw) -> 4x4 pose (identity rotation for a null
    quaternion)."""
    o = 1 if has_timestamp else 0
    T = np.identity(4)
    T[:3, 3] = scale_translation * np.array(l[o:o + 3], dtype=np.float64)
    q = np.array(l[o + 3:o + 7], dtype=np.float64)
    nq = np.dot(q, q)
    if nq >= _EPS:
        x, y, z, w = q * np.sqrt(2.0 / nq)
        T[:3, :3] = _rotation_from_scaled_quaternion(w, x, y, z)
    return T


def rpe_translation_metric(T):
    """transformations.py:2078"""
    return np.linalg.norm(np.asarray(T)[:3, 3])


def rpe_rotation_metric(T):
    """transformations.py:2096: rotation angle of the 3x3 block."""
    d = 0.5 * (np.trace(np.asarray(T)[0:3, 0:3]) - 1.0)
    return np.arccos(min(1.0, max(-1.0, d)))


def rpe(Ta, Tb):
    """transformations.py:2029: relative pose error inv(Ta) . Tb."""
    return np.dot(inverse_matrix(Ta), Tb)
