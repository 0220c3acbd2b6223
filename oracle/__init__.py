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
        _lib = ctypes.CDLL(LIB_PATH)
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


def sort_matches(keys):
    """keys [nq] or [nq,1] u32 -> order [nq] i32, stable by distance."""
    keys = _c(keys, np.uint32).reshape(-1)
    order = np.empty(keys.shape[0], dtype=np.int32)
    lib().orc_sort_matches(_p(keys), ctypes.c_int32(keys.shape[0]), _p(order))
    return order
