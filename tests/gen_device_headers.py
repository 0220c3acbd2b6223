"""Device headers that are the TEXT of an oracle header (same arithmetic on both sides, by construction):
    oracle/trig_core.h -> vo_single_camera_sos_amd/csrc/trig_core.h
    oracle/gp3p_core.h -> vo_single_camera_sos_amd/csrc/gp3p_core.h
(orc_ -> sv_, ORC_ -> SV_, static inline -> __device__ static, the oracle's provenance header replaced by a notice).
Run after editing an oracle header:   python tests/gen_device_headers.py   (tests/test_abi.py checks that it was)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [("oracle/trig_core.h", "vo_single_camera_sos_amd/csrc/trig_core.h"),
         ("oracle/gp3p_core.h", "vo_single_camera_sos_amd/csrc/gp3p_core.h")]

NOTICE = {
    "oracle/trig_core.h":
        "// sin / cos / atan in double precision from + - * / and comparisons only (Cody-Waite reduction, minimax\n"
        "// polynomials): the panorama geometry of the device code.  The CPU oracle evaluates the SAME text (%s, with\n"
        "// its own prefix) so that both sides agree to the bit; tests/gen_device_headers.py keeps the two files identical\n"
        "// and tests/test_abi.py checks it.  Edit both through that script.",
    "oracle/gp3p_core.h":
        "// Generalised P3P, the minimal solver of the non-central absolute-pose RANSAC (the reference's\n"
        "// absolute_pose_noncentral_ransac \"will ALWAYS use GP3P\", omnistereo/pose_est_tools.py:696, :785): three quadrics in\n"
        "// the three depths -> an octic in the first one (resultants) -> Laguerre roots -> Newton polish -> triangle alignment;\n"
        "// up to 8 poses, the fourth correspondence picks one.  GENERATED from %s by tests/gen_device_headers.py (same text,\n"
        "// device prefixes): the CPU oracle evaluates the same operations in the same order, tests/test_abi.py checks that the\n"
        "// two files stay identical.  The derivation and the independent checks are documented in the oracle header.",
}


def device_text(src_text, src_name):
    t = src_text
    t = re.sub(r"/\* TEST INFRASTRUCTURE.*?\*/", lambda m: NOTICE[src_name] % src_name, t, count=1, flags=re.S)
    t = t.replace("#pragma once\n", "#pragma once\n#include <hip/hip_runtime.h>\n", 1)
    t = t.replace("static inline", "__device__ static")
    t = re.sub(r"\borc_", "sv_", t)
    t = re.sub(r"\bORC_", "SV_", t)
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
