"""Build-time guard on the matcher's ISA (VERDICT / ADVICE round 3: the MFMA matcher once returned wrong keys under load).
scripts/check_mfma_hazards.py rebuilds the control-flow graph of every kernel with a v_mfma and measures, on EVERY path, the
wait states between an MFMA and the first touch of its result registers -- independently of the compiler's hazard
recogniser, which DESIGN.md section 13 shows to be unsound across the diamonds that conditional MFMAs create.
No GPU needed: hipcc cross-compiles to assembly."""
import glob
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_mfma_hazards as chk  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"


def test_checker_flags_the_conditional_mfma_form():
    """The <2,1> kernel of the round-3 matcher with a wave-uniform `if` around every MFMA and around the epilogue
    (profiles/round4/mfma_branchy_form.diff), compiled by ROCm 7.2 hipcc in VGPR form: on the path a wave takes when its
    second query tile is dead the first accumulator of tile 0 is read 6 wait states after the last MFMA -- 12 are owed."""
    text = open(os.path.join(ROOT, "tests", "golden", "mfma_branchy_2_1_vgpr_form.s")).read()
    report, bad = chk.check_text(text)
    assert bad and len(report) == 1
    r = next(iter(report.values()))
    assert r["mfma"] == 16 and r["branches_inside_chains"] == 14
    assert r["violations"] and r["min_wait"] == 6, r


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_library_mfma_kernels_keep_their_wait_states():
    """Every .hip of the library that issues an MFMA, compiled with csrc/Makefile's flags: accumulators stay in VGPRs (no
    v_accvgpr moves: -mllvm -amdgpu-mfma-vgpr-form still does what match.hip relies on), no conditional branch inside an
    accumulation chain, and no touch of an MFMA result earlier than passes + 4 wait states on any path."""
    makefile = open(os.path.join(ROOT, "vo_single_camera_sos_amd", "csrc", "Makefile")).read()
    assert "-amdgpu-mfma-vgpr-form" in makefile and "-ffp-contract=off" in makefile
    srcs = [p for p in sorted(glob.glob(os.path.join(ROOT, "vo_single_camera_sos_amd", "csrc", "*.hip")))
            if "__builtin_amdgcn_mfma" in open(p).read()]
    assert srcs, "the matcher is expected to use the matrix cores"
    seen = 0
    for src in srcs:
        report, bad = chk.check_text(chk.compile_hip(src))
        for sym, r in report.items():
            seen += 1
            assert r["accvgpr"] == 0, (sym, r["accvgpr"])
            assert r["branches_inside_chains"] == 0, sym
            assert not r["violations"], (sym, r["violations"][:3])
        assert not bad
    assert seen >= 4   # match_hamming_mfma_kernel<1|2, 1|2>
