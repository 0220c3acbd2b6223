"""Device headers that are the TEXT of an oracle header (same arithmetic on both sides, by construction):
    oracle/trig_core.h -> vo_single_camera_sos_amd/csrc/trig_core.h
    oracle/gp3p_core.h -> vo_single_camera_sos_amd/csrc/gp3p_core.h
    oracle/ransac_core.h -> vo_single_camera_sos_amd/csrc/ransac_core.h     (forced inlining on the device)
    oracle/relpose_core.h -> vo_single_camera_sos_amd/csrc/relpose_core.h
    oracle/epnp_core.h -> vo_single_camera_sos_amd/csrc/epnp_core.h         (minus the "@oracle-only" section; the device's
                                                                            register-resident 12 x 12 eigen-solver is the
                                                                            hand-written csrc/epnp_eig12_reg.h)
(orc_ -> sv_, ORC_ -> SV_, static inline -> __device__ static, the oracle's provenance header replaced by a notice).
Run after editing an oracle header:   python tests/gen_device_headers.py   (tests/test_abi.py checks that it was)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [("oracle/trig_core.h", "vo_single_camera_sos_amd/csrc/trig_core.h"),
         ("oracle/gp3p_core.h", "vo_single_camera_sos_amd/csrc/gp3p_core.h"),
         ("oracle/ransac_core.h", "vo_single_camera_sos_amd/csrc/ransac_core.h"),
         ("oracle/epnp_core.h", "vo_single_camera_sos_amd/csrc/epnp_core.h"),
         ("oracle/relpose_core.h", "vo_single_camera_sos_amd/csrc/relpose_core.h")]
# device-only continuation appended to the generated text (hand-written, see the file's header)
TAIL = {"oracle/epnp_core.h": '\n#include "epnp_eig12_reg.h"\n'}
FORCE_INLINE = {"oracle/ransac_core.h"}

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
    "oracle/epnp_core.h":
        "// Device-side EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009) for the central absolute-pose RANSAC with algorithm\n"
        "// \"EPNP\" (omnistereo/pose_est_tools.py:697, :915: OpenGV solves 6-point samples with EPnP): control points,\n"
        "// barycentric coordinates, [the null space of M^T M: epnp_eig12_reg.h], the three beta initialisations with five\n"
        "// Gauss-Newton steps each, absolute orientation, the candidate with the smallest reprojection error; plus the sampler\n"
        "// of k distinct indices.  One lane per hypothesis, every array index a compile-time constant after unrolling.\n"
        "// GENERATED from %s by tests/gen_device_headers.py (the oracle's text minus its \"oracle only\" section, device\n"
        "// prefixes): both sides evaluate the same operations in the same order; tests/test_abi.py checks the two files.",
    "oracle/relpose_core.h":
        "// Device-side numeric core of the 2D-2D relative-pose RANSAC (pyopengv.relative_pose_ransac, reference call site\n"
        "// omnistereo/pose_est_tools.py:78): the reference's own score of a correspondence under a relative pose\n"
        "// (pose_est_tools.py:150-203), the decomposition of an essential matrix, the minimal solvers, one hypothesis per lane.\n"
        "// GENERATED from %s by tests/gen_device_headers.py (same text, device prefixes): the CPU oracle evaluates the same\n"
        "// operations in the same order, tests/test_abi.py checks that the two files stay identical.",
    "oracle/ransac_core.h":
        "// Device-side numeric core of the absolute-pose RANSAC (K8/K10) and LM refinement (K9): counter-based sampler, Kneip\n"
        "// P3P, real quartic roots, score, Cayley parametrisation, LM pieces.  Reference call sites replaced:\n"
        "// pyopengv.absolute_pose_noncentral_ransac (omnistereo/pose_est_tools.py:785), absolute_pose_ransac (:915),\n"
        "// *_optimize_nonlinear (:830, :937); score definition from pose_est_tools.py:150-203 (+ :181-185 non-central).\n"
        "// Only + - * / sqrt and comparisons, fully parenthesised, built with -ffp-contract=off.  GENERATED from %s by\n"
        "// tests/gen_device_headers.py (same text, device prefixes): the CPU oracle evaluates the same operations in the same\n"
        "// order, tests/test_abi.py checks that the two files stay identical.",
}


def device_text(src_text, src_name):
    t = src_text
    t = re.sub(r"\n/\* @oracle-only: begin.*?/\* @oracle-only: end \*/\n", "\n", t, flags=re.S)
    t = re.sub(r"/\* TEST INFRASTRUCTURE.*?\*/", lambda m: NOTICE[src_name] % src_name, t, count=1, flags=re.S)
    t = t.replace("#pragma once\n", "#pragma once\n#include <hip/hip_runtime.h>\n", 1)
    t = t.replace("static inline", "__device__ __forceinline__ static" if src_name in FORCE_INLINE else "__device__ static")
    t = re.sub(r"\borc_", "sv_", t)
    t = re.sub(r"\bORC_", "SV_", t)
    return t + TAIL.get(src_name, "")


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
