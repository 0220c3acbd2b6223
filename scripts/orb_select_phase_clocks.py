"""Phase clocks of orb_select_kernel under the ORB bench, one stream of 256 pairs.  Needs the DEBUG-TIMING build of the library
(`make -C vo_single_camera_sos_amd/csrc EXTRA=-DSOSVO_DEBUG_TIMING` after touching orb.hip; rebuild without it afterwards)."""
import ctypes, os, sys, subprocess, runpy, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vo_single_camera_sos_amd import _lib
lib = _lib.load()
fn = lib.sosvo_debug_orb_select_ticks
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int32]
def read(reset=1):
    a = (ctypes.c_ulonglong * 8)()
    fn(a, reset)
    return list(a)
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-h2d", "--no-isolated", "--no-sub", "--streams", "1", "--pairs-per-gpu", "256",
            "--detector", "ORB", "--median-win-size", "0", "--features-per-mask", "230"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
    except SystemExit:
        pass
t = read()
nprob = 1024 * 12 * 5  # problems x (warmup + steps)
names = ["walk+hist", "threshold+compact", "harris", "sort", "retain+orientation+out"]
tot = sum(t[:5])
for k, n in enumerate(names):
    print("%-26s %7.2f us per problem (all levels)  %5.1f %%" % (n, t[k] / nprob / 100.0, 100.0 * t[k] / tot))
print("sum %.2f us per problem" % (tot / nprob / 100.0))
