"""Host-side mirror of the one hot-path helper of omnistereo/common_cv.py.  numpy, as in the reference (the
batched pipeline applies the same gates inside the stereo / frame-to-frame assembly kernels)."""
import numpy as np


def filter_pixel_correspondences(matched_points_top, matched_points_bot, min_rectified_disparity, max_horizontal_diff):
    """common_cv.py:167-188: |u_top - u_bot| <= max_horizontal_diff (if > 0) and v_top - v_bot >=
    min_rectified_disparity (if >= 0), both limits inclusive -> bool array."""
    matched_points_top = np.asarray(matched_points_top)
    matched_points_bot = np.asarray(matched_points_bot)
    if max_horizontal_diff > 0:
        validation_hor_diff = np.abs(matched_points_top[..., 0] - matched_points_bot[..., 0]) <= max_horizontal_diff
    else:
        validation_hor_diff = np.ones(shape=(matched_points_top.shape[:-1]), dtype="bool")
    if min_rectified_disparity >= 0:
        validation_min_disparity = matched_points_top[..., 1] - matched_points_bot[..., 1] >= min_rectified_disparity
        return np.logical_and(validation_hor_diff, validation_min_disparity)
    return validation_hor_diff
