"""Host-side mirror of the helpers of omnistereo/common_tools.py that the VO loop uses: unit conversion, the
TUM pose-file reader / writer, path creation, attribute copying (keyframes).  Pinned by tests/golden/transforms.npz."""
import errno
import os

import numpy as np

from . import transformations as tr

_UNIT_IN_M = {"mm": 0.001, "cm": 0.01, "m": 1.0}
_EXACT = {("cm", "mm"): 10.0, ("cm", "m"): 0.01, ("mm", "cm"): 0.1, ("mm", "m"): 0.001, ("m", "mm"): 1000.0,
          ("m", "cm"): 100.0}


def get_length_units_conversion_factor(input_units, output_units):
    """common_tools.py:580-597 (the same literal factors; unknown pairs -> 1.0)."""
    return _EXACT.get((input_units, output_units), 1.0)


def make_sure_path_exists(path):
    """common_tools.py:86-92"""
    try:
        os.makedirs(path)
    except OSError as exception:
        if exception.errno != errno.EEXIST:
            raise


def str2bool(v):
    return str(v).lower() in ("yes", "true", "t", "1")


def copy_only_attributes(objfrom, objto, exclude_names=()):
    """common_tools.py:154-159: every non-dunder member of objfrom is set on objto."""
    for n in dir(objfrom):
        if "__" in n or n in exclude_names:
            continue
        v = getattr(objfrom, n)
        if callable(v) and hasattr(type(objto), n):
            continue  # methods come with the class
        setattr(objto, n, v)


def save_as_tum_poses_to_file(output_tum_filename, poses_7_list, input_units, input_format, output_units="m"):
    """common_tools.py:599-621: one line per pose, the line number as time stamp; input_format "tr" =
    [qw, qx, qy, qz, tx, ty, tz], "tum" = [tx, ty, tz, qx, qy, qz, qw]."""
    k = get_length_units_conversion_factor(input_units, output_units)
    with open(output_tum_filename, "w") as f:
        for stamp, pose in enumerate(poses_7_list):
            if "tr" in input_format.lower():
                qw, qi, qj, qk = pose[:4]
                tx, ty, tz = pose[4:]
            else:
                tx, ty, tz = pose[:3]
                qi, qj, qk, qw = pose[3:]
            print("%d %.9f %.9f %.9f %.9f %.9f %.9f %.9f" % (stamp, k * tx, k * ty, k * tz, qi, qj, qk, qw), file=f)


def get_poses_from_file(poses_filename, input_units="m", output_working_units="m", indices=None, pose_format="tum",
                        zero_up_wrt_origin=False, initial_T=None, delimiter=None):
    """common_tools.py:623-736.  pose_format "tum": stamp tx ty tz qx qy qz qw per line, blank-separated; "povray":
    tx, ty, tz, rot_x, rot_y, rot_z per line, comma-separated, angles in degrees about the static x, y, z axes in that
    order ('#' comments in both) -> (list of [tx, ty, tz, qx, qy, qz, qw] in output units, list of 4x4 matrices).
    With zero_up_wrt_origin the first valid pose becomes the identity (every pose is pre-multiplied by its inverse,
    `initial_T` post-multiplied); rows holding NaN give NaN entries."""
    fmt = pose_format.lower()
    if fmt not in ("tum", "povray"):
        raise ValueError("pose_format %r: 'tum' or 'povray'" % pose_format)
    if delimiter is None and fmt == "povray":
        delimiter = ","
    grid = np.loadtxt(poses_filename, delimiter=delimiter if delimiter and delimiter != " " else None,
                      usecols=tuple(range(8 if fmt == "tum" else 6)), comments="#", ndmin=2)
    if len(grid) == 0:
        raise ValueError("no poses in %s" % poses_filename)
    k = get_length_units_conversion_factor(input_units, output_working_units)
    if indices is None or len(indices) == 0:
        indices = range(len(grid))
    n = len(indices)
    poses7, mats = n * [None], n * [None]
    T_offset, have_offset = tr.identity_matrix(), False
    apply_init = initial_T is not None
    if initial_T is None:
        initial_T = tr.identity_matrix()
    for pose_number in indices:
        row = grid[pose_number]
        if np.any(np.isnan(row)):
            entry = 7 * [np.nan]
        elif fmt == "povray":
            qw, qx, qy, qz = tr.quaternion_from_euler(np.deg2rad(float(row[3])), np.deg2rad(float(row[4])),
                                                      np.deg2rad(float(row[5])), "sxyz")
            entry = [k * float(row[0]), k * float(row[1]), k * float(row[2]), qx, qy, qz, qw]
        else:
            entry = [k * float(row[1]), k * float(row[2]), k * float(row[3]), float(row[4]), float(row[5]), float(row[6]),
                     float(row[7])]
        poses7[pose_number] = entry
        T = tr.transform44_from_TUM_entry(entry, scale_translation=1.0, has_timestamp=False)
        mats[pose_number] = T
        if (zero_up_wrt_origin and not np.any(np.isnan(T))) or apply_init:
            if zero_up_wrt_origin and not have_offset:
                T_offset = tr.inverse_matrix(T)
                have_offset = True
            T = tr.concatenate_matrices(T_offset, T, initial_T)
            mats[pose_number] = T
            e = 7 * [0.0]
            if np.any(np.isnan(T)):
                e[3:] = 4 * [np.nan]
            else:
                e[6], e[3], e[4], e[5] = tr.quaternion_from_matrix(T, isprecise=False)
            e[:3] = tr.translation_from_matrix(T)
            poses7[pose_number] = e
    return poses7, mats
