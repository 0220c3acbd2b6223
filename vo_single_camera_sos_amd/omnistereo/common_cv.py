"""Host-side mirror of the helpers of omnistereo/common_cv.py on either side of the hot path: the pixel gates
(numpy, as in the reference; the batched pipeline applies the same gates inside the assembly kernels) and the
sequence readers of the VO loop (image list, BGR image, 16-bit depth PNG) -- on Pillow, cv2 does not exist here."""
import fnmatch
import os

import numpy as np


def get_images(filename_template, indices_list=(), show_images=False, return_names_only=False):
    """common_cv.py:1293-1352: the files matching `filename_template` (directory + fnmatch pattern), optionally
    sub-selected by position.  The listing is SORTED here (the reference takes os.listdir order, which is
    arbitrary)."""
    path_to_files, pattern = os.path.split(filename_template)
    names = sorted(fnmatch.filter(os.listdir(path_to_files), pattern))
    all_names = [os.path.join(path_to_files, n) for n in names]
    if indices_list is None or len(indices_list) == 0:
        indices_list = range(len(all_names))
    selected = [all_names[i] for i in indices_list]
    if return_names_only:
        return selected
    return [imread(fn) for fn in selected]


def imread(filename):
    """cv2.imread(filename): uint8 [rows, cols, 3] in B, G, R order (None when unreadable)."""
    from PIL import Image
    try:
        with Image.open(filename) as im:
            rgb = np.asarray(im.convert("RGB"))
    except (OSError, ValueError):
        return None
    return np.ascontiguousarray(rgb[..., ::-1])


def imwrite(filename, bgr_or_gray):
    from PIL import Image
    a = np.asarray(bgr_or_gray)
    Image.fromarray(a[..., ::-1] if a.ndim == 3 else a).save(filename)


def get_depthmap_float32_from_png(depth_img_filename, conversion_factor=1.0):
    """common_cv.py:2150-2158: the 16-bit depth PNG times the conversion factor, as float32."""
    from PIL import Image
    with Image.open(depth_img_filename) as im:
        raw = np.asarray(im)
    return np.float32(conversion_factor * raw)


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
