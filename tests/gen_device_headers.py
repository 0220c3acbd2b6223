"""Device headers that are the TEXT of an oracle header (same arithmetic on both sides, by construction):
    oracle/trig_core.h -> vo_single_camera_sos_amd/csrc/trig_core.h   (orc_ -> sv_, static inline -> __device__ static)
Run after editing the oracle header:   python tests/gen_device_headers.py   (tests/test_abi.py checks that it was)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [("oracle/trig_core.h", "vo_single_camera_sos_amd/csrc/trig_core.h")]


def device_text(src_text, src_name):
    t = src_text
    t = re.sub(r"/\* TEST INFRASTRUCTURE.*?\*/", "// sin / cos / atan in double precision from + - * / and comparisons only (Cody-Waite reduction, minimax\n"
               "// polynomials): the panorama geometry of the device code.  The CPU oracle evaluates the SAME text (%s, with\n"
               "// its own prefix) so that both sides agree to the bit; tests/gen_device_headers.py keeps the two files identical\n"
               "// and tests/test_abi.py checks it.  Edit both through that script." % src_name, t, count=1, flags=re.S)
    t = t.replace("#pragma once\n", "#pragma once\n#include <hip/hip_runtime.h>\n", 1)
    t = t.replace("static inline", "__device__ static")
    t = re.sub(r"\borc_", "sv_", t)
    return t


def main(check=False):
    ok = True
    for src, dst in PAIRS:
        want = device_text(open(os.path.join(ROOT, src)).read(), src)
        path = os.path.join(ROOT, dst)
        if check:
            ok = ok and os.path.exists(path) and open(path).read() == want
        else:
            with open(path, "w") as f:
                f.write(want)
    return ok


if __name__ == "__main__":
    main()
