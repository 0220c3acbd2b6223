"""The 256 BRIEF test pairs used by libsosvo's descriptor kernels (K6).

Default (round 4): OpenCV's learned table `bit_pattern_31_` -- what `cv2.ORB_create(...).compute` uses at the
reference's call sites (omnistereo/camera_models.py:1682, :1765; pose_est_tools.py:520, :553).  It is third-party DATA
(3-clause BSD), shipped as data/orb_bit_pattern_31.txt (taken from scikit-image's copy; the file's header says how) and
also exported by the C ABI (`sosvo_orb_bit_pattern_31`).  |coordinate| <= 13: rotated by any angle a test point stays
within 19 px of the keypoint, inside the 31-px border ORB.compute enforces.

`seeded_pattern()` is the table rounds 1-3 used while OpenCV's was thought unobtainable: 512 points, each coordinate
the sum of three uniform draws in [-4, 4] from a splitmix64 stream with a fixed seed (|coordinate| <= 12)."""
import os

import numpy as np

_MASK = (1 << 64) - 1


def _splitmix_stream(seed):
    state = seed
    while True:
        state = (state + 0x9E3779B97F4A7C15) & _MASK
        z = state
        z ^= z >> 30
        z = (z * 0xBF58476D1CE4E5B9) & _MASK
        z ^= z >> 27
        z = (z * 0x94D049BB133111EB) & _MASK
        z ^= z >> 31
        yield z


_OPENCV = None


def opencv_pattern():
    """-> int8 array [512, 2]: OpenCV's bit_pattern_31_; test t compares points 2t and 2t+1."""
    global _OPENCV
    if _OPENCV is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "orb_bit_pattern_31.txt")
        rows = np.loadtxt(path, dtype=np.int64, comments="#")
        if rows.shape != (256, 4) or np.abs(rows).max() > 13:
            raise ValueError("%s: not the 256 x 4 table" % path)
        _OPENCV = np.ascontiguousarray(rows.reshape(512, 2).astype(np.int8))
    return _OPENCV.copy()


def orb_pattern(kind=None):
    """The descriptor pattern the package hands to libsosvo: "opencv" (default) or "seeded"; with kind None the environment
    variable SOSVO_ORB_PATTERN chooses (unset: "opencv")."""
    if kind is None:
        kind = os.environ.get("SOSVO_ORB_PATTERN", "opencv")
    if kind == "opencv":
        return opencv_pattern()
    if kind == "seeded":
        return seeded_pattern()
    raise ValueError("unknown pattern %r" % (kind,))


def seeded_pattern():
    """-> int8 array [512, 2]: the seeded table of rounds 1-3; test t compares points 2t and 2t+1."""
    gen = _splitmix_stream(0x0B5EED5EED5EED01)
    flat = np.empty(1024, dtype=np.int8)
    for i in range(1024):
        v = 0
        for _ in range(3):
            v += int((next(gen) >> 32) % 9) - 4
        flat[i] = v
    for t in range(256):  # a test whose two points coincide carries no information
        if flat[4 * t] == flat[4 * t + 2] and flat[4 * t + 1] == flat[4 * t + 3]:
            flat[4 * t + 2] += 1 if flat[4 * t + 2] < 12 else -1
    return flat.reshape(512, 2)


def angle_cos_sin(angle_degrees):
    """float32 (cos, sin) of a keypoint angle the way ORB's descriptor stage evaluates them: the angle is
    scaled to radians in float32, cos/sin are taken in double and rounded to float32."""
    a = np.float32(angle_degrees) * np.float32(np.pi / 180.0)
    return np.float32(np.cos(np.float64(a))), np.float32(np.sin(np.float64(a)))


GFT_KEYPOINT_ANGLE = -1.0  # cv2.KeyPoint_convert leaves angle = -1 (camera_models.py:1752)
