"""ctypes binding of libsosvo.so (the C ABI declared in include/sosvo.h).

The product path has no CPU fallback: if the HIP library is missing or a symbol is
absent this module raises, it never routes anywhere else.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsosvo.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sosvo.h")

c_i32 = ctypes.c_int32
c_f64 = ctypes.c_double
c_f32 = ctypes.c_float
c_u64 = ctypes.c_uint64
c_p = ctypes.c_void_p

SOSVO_OK = 0
STATUS_NAMES = {0: "SOSVO_OK", -1: "SOSVO_ERR_ARG", -2: "SOSVO_ERR_HIP", -3: "SOSVO_ERR_NODEVICE",
                -4: "SOSVO_ERR_CAPACITY"}

KEY_SHIFT = 20
KEY_IDX_MASK = 0xFFFFF
KEY_NONE = 0xFFFFFFFF
DESC_BYTES = 32


class SosvoError(RuntimeError):
    pass


# name -> (restype, argtypes); must list every function of include/sosvo.h (tested).
SIGNATURES = {
    "sosvo_abi_version": (c_i32, []),
    "sosvo_orb_bit_pattern_31": (c_i32, [c_p]),
    "sosvo_create": (c_i32, [ctypes.POINTER(c_p), c_i32, c_p]),
    "sosvo_destroy": (c_i32, [c_p]),
    "sosvo_set_stream": (c_i32, [c_p, c_p]),
    "sosvo_set_hint": (c_i32, [c_p, c_i32, c_i32]),
    "sosvo_synchronize": (c_i32, [c_p]),
    "sosvo_last_error": (ctypes.c_char_p, [c_p]),
    "sosvo_timer_start": (c_i32, [c_p]),
    "sosvo_timer_stop": (c_i32, [c_p]),
    "sosvo_timer_elapsed_ms": (c_i32, [c_p, ctypes.POINTER(c_f32)]),
    "sosvo_profile_enable": (c_i32, [c_p, c_i32]),
    "sosvo_profile_count": (c_i32, [c_p]),
    "sosvo_debug_fill_scratch": (c_i32, [c_p, c_i32]),
    "sosvo_profile_get": (c_i32, [c_p, c_i32, ctypes.c_char_p, c_i32, ctypes.POINTER(c_f32)]),
    "sosvo_unwrap": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sosvo_unwrap_prepare": (c_i32, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sosvo_unwrap_table": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sosvo_median_gray": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sosvo_unwrap_median_gray": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sosvo_unwrap_median_gray_rows": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p]),
    "sosvo_gray_rows_needed": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_f32, c_f32, c_p]),
    "sosvo_detect_gft": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_f64, c_f64, c_i32, c_i32, c_p,
                                 c_p, c_p]),
    "sosvo_describe_orb": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_f32, c_f32, c_p, c_i32,
                                   c_p]),
    "sosvo_describe_orb_rows": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_f32, c_f32, c_p, c_i32,
                                        c_p, c_p]),
    "sosvo_orb_pyramid_pixels": (ctypes.c_int64, [c_i32, c_i32]),
    "sosvo_orb_mask_pyramid": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sosvo_detect_orb": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    "sosvo_describe_orb_levels": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p]),
    "sosvo_detect_describe_orb": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p,
                                          c_p, c_p]),
    "sosvo_detect_fast": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    "sosvo_detect_agast": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    "sosvo_match_hamming": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sosvo_match_radius": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p]),
    "sosvo_sort_matches": (c_i32, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_p]),
    "sosvo_pano_to_bearing": (c_i32, [c_p, c_p, c_i32, c_f64, c_f64, c_f64, c_f64, c_p, c_p, c_p]),
    "sosvo_triangulate_midpoint": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_i32, c_p, c_p, c_p]),
    "sosvo_triangulate2": (c_i32, [c_p, c_p, c_p, c_i32, c_p, c_p, c_p]),
    "sosvo_range_filter": (c_i32, [c_p, c_p, c_i32, c_f64, c_f64, c_p]),
    "sosvo_rgbd_backproject": (c_i32, [c_p, c_p, c_i32, c_i32, c_p, c_p, c_i32, c_f64, c_f64, c_f64, c_f64, c_f64,
                                       c_i32, c_p, c_p]),
    "sosvo_stereo_assemble": (c_i32, [c_p] + [c_p] * 9 + [c_i32] * 4 + [c_p] * 9),
    "sosvo_f2f_assemble": (c_i32, [c_p] + [c_p] * 7 + [c_i32] + [c_p] * 6 + [c_i32, c_i32] + [c_p] * 7),
    "sosvo_ransac_abs_pose": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_p, c_i32, c_i32, c_f64,
                                      c_i32, c_i32, c_u64, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sosvo_match_l2": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sosvo_ransac_rel_pose": (c_i32, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_f64, c_i32, c_i32, c_u64, c_p, c_p, c_p, c_p,
                                      c_p, c_p]),
    "sosvo_refine_abs_pose": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_p, c_i32, c_i32, c_p, c_p, c_i32,
                                      c_p, c_p, c_p]),
    "sosvo_rgbd_assemble": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p]),
    "sosvo_f2f_assemble_central": (c_i32, [c_p, c_f64, c_f64, c_p, c_p, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_i32, c_i32,
                                           c_p, c_p, c_p, c_p, c_p]),
    "sosvo_frame_pair_batch_workspace": (ctypes.c_size_t, [c_p]),
    "sosvo_frame_pair_batch": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    "sosvo_frame_pair_batch_streams_workspace": (ctypes.c_size_t, [c_p, c_i32]),
    "sosvo_frame_pair_batch_streams": (c_i32, [c_p, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    "sosvo_frame_pair_batch_streams_enqueue": (c_i32, [c_p, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    "sosvo_frame_pair_batch_streams_join": (c_i32, [c_p]),
    "sosvo_sequence_workspace": (ctypes.c_size_t, [c_p, c_i32, c_i32]),
    "sosvo_sequence_front_end": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p, ctypes.c_size_t]),
    "sosvo_sequence_track": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_p, c_p, c_i32, c_u64, c_p, ctypes.c_size_t, c_p]),
    "sosvo_sequence_copy_slot": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, ctypes.c_size_t]),
    "sosvo_sequence_frame_counts": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, ctypes.c_size_t, c_p]),
    "sosvo_rgbd_sequence_workspace": (ctypes.c_size_t, [c_p, c_i32, c_i32]),
    "sosvo_rgbd_sequence_front_end": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p, ctypes.c_size_t]),
    "sosvo_rgbd_sequence_track": (c_i32, [c_p, c_p, c_i32, c_i32, c_p, c_p, c_i32, c_u64, c_p, ctypes.c_size_t, c_p]),
    "sosvo_rgbd_sequence_copy_slot": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, ctypes.c_size_t]),
    "sosvo_rgbd_sequence_frame_counts": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p, ctypes.c_size_t, c_p]),
    "sosvo_rgbd_pair_batch_workspace": (ctypes.c_size_t, [c_p]),
    "sosvo_rgbd_pair_batch": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
}

HINT_SHARED_DEVICE = 1
HINT_SCORE_FP64_ONLY = 2
FLAG_CAM_ROT_IDENTITY = 1
FLAG_EPNP = 2
FLAG_GP3P = 4
FLAG_TWOPT = 8
REL_FIVEPT = 5
REL_SEVENPT = 7
REL_EIGHTPT = 8


class Rig(ctypes.Structure):
    """Mirror of `struct sosvo_rig` (include/sosvo.h)."""
    _fields_ = [("pano_top", c_f64 * 4), ("pano_bot", c_f64 * 4), ("F_top", c_f64 * 3), ("F_bot", c_f64 * 3),
                ("min_range", c_f64), ("max_range", c_f64), ("stereo_min_disp", c_f64),
                ("stereo_max_hdiff", c_f64), ("f2f_max_hdiff", c_f64), ("pct_good_matches", c_f64)]



class RgbdCam(ctypes.Structure):
    """Mirror of `struct sosvo_rgbd_cam` (include/sosvo.h)."""
    _fields_ = [("fx", c_f64), ("fy", c_f64), ("cx", c_f64), ("cy", c_f64), ("focal_length_m", c_f64),
                ("depth_is_Z", c_i32), ("reserved", c_i32), ("min_range", c_f64), ("max_range", c_f64)]


class BatchCfg(ctypes.Structure):
    """Mirror of `struct sosvo_batch_cfg` (include/sosvo.h)."""
    _fields_ = [("n_pairs", c_i32), ("H", c_i32), ("W", c_i32), ("rows", c_i32), ("cols", c_i32), ("nmask", c_i32),
                ("kp_cap", c_i32), ("frame_cap", c_i32), ("median_ksize", c_i32), ("max_corners", c_i32), ("edge", c_i32),
                ("ransac_max_iter", c_i32), ("ransac_adaptive", c_i32), ("lm_max_iter", c_i32), ("quality", c_f64),
                ("min_distance", c_f64), ("ransac_threshold", c_f64), ("seed", c_u64), ("cos_a", c_f32), ("sin_a", c_f32),
                ("ransac_flags", c_i32), ("reserved", c_i32)]


class RgbdBatchCfg(ctypes.Structure):
    """Mirror of `struct sosvo_rgbd_batch_cfg` (include/sosvo.h)."""
    _fields_ = [("n_pairs", c_i32), ("rows", c_i32), ("cols", c_i32), ("kp_cap", c_i32), ("frame_cap", c_i32),
                ("median_ksize", c_i32), ("max_corners", c_i32), ("edge", c_i32), ("ransac_max_iter", c_i32),
                ("ransac_adaptive", c_i32), ("lm_max_iter", c_i32), ("flags", c_i32), ("quality", c_f64),
                ("min_distance", c_f64), ("ransac_threshold", c_f64), ("pct_good_matches", c_f64), ("f2f_max_hdiff", c_f64),
                ("seed", c_u64), ("cos_a", c_f32), ("sin_a", c_f32)]


_lib = None


def declared_functions(header_path=HEADER_PATH):
    """Names of all functions declared in include/sosvo.h."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sosvo_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load libsosvo.so (once).  Raises SosvoError loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SosvoError(
            "libsosvo.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vo_single_camera_sos_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # torch must be imported first so that its bundled libamdhip64.so.7 is the HIP runtime
    # this library binds to (one runtime per process).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise SosvoError("libsosvo.so does not export %s; rebuild it" % name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
