"""The 256 BRIEF test pairs used by libsosvo's descriptor kernel (K6).

OpenCV's ORB ships a learned 256-pair table (`bit_pattern_31_`); that table is data of a third-party
library that is not in the reference tree and cannot be obtained offline, so this package defines its
own fixed pattern: 512 points (x, y), each coordinate the sum of three uniform draws in [-4, 4]
(bell-shaped as the BRIEF paper recommends, |coordinate| <= 12 so the patch rotated by any angle stays
inside the 31-px border that ORB.compute enforces), drawn from a splitmix64 stream with a fixed seed.
Descriptors are therefore comparable among themselves (all the VO path needs) but not with OpenCV's."""
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


def orb_pattern():
    """-> int8 array [512, 2]: test t compares points 2t and 2t+1."""
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
