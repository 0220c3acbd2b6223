import sys, ctypes, subprocess, json
sys.argv = ["bench.py", "--steps", "5", "--warmup", "1", "--no-cpu"]
import torch
from vo_single_camera_sos_amd import _lib
lib = _lib.load()
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
out = (ctypes.c_ulonglong * 8)()
lib.sosvo_debug_counters(out)
v = list(out)
print("DBG", v, "per-problem cycles p1,p2,p3:", [x / max(v[5],1) for x in v[:3]], "avg cand", v[4] / max(v[5],1))
