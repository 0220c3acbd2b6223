"""Build-time guard on ransac_score_kernel's tier-1 form: what the 0.75 -> 0.47 ms rests on is visible
in the assembly -- 128 VGPRs (four waves per SIMD), no packed single-precision instructions (the SLP vectoriser, switched off
for this file in csrc/Makefile, packs pairs of chains with a move per broadcast operand and spills inside the loop; the
hand-packed form measured slower too), and no scratch access inside a loop (spills of the one-time set-up are fine).
No GPU needed: hipcc cross-compiles to assembly."""
import os
import re
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_mfma_hazards as chk  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_tier1_kernel_keeps_its_shape():
    makefile = open(os.path.join(ROOT, "vo_single_camera_sos_amd", "csrc", "Makefile")).read()
    m = re.search(r"^FLAGS_ransac\s*:=\s*(.*)$", makefile, re.M)
    assert m and "-fno-slp-vectorize" in m.group(1)
    text = chk.compile_hip(os.path.join(ROOT, "vo_single_camera_sos_amd", "csrc", "ransac.hip"), extra=m.group(1).split())
    # the tier-1 instance: ransac_score_kernel<true, 4, true>
    syms = re.findall(r"^(_ZN\S*ransac_score_kernelILb1ELi4ELb1E\S*):", text, re.M)
    assert len(set(syms)) == 1, syms
    sym = syms[0]
    body = text[text.index(sym + ":"):]
    body = body[:body.index(".end_amdhsa_kernel")]
    vg = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", body).group(1))
    assert vg <= 128, "tier-1 scoring kernel at %d VGPRs: fewer than four waves per SIMD" % vg
    assert "v_pk_fma_f32" not in body and "v_pk_mul_f32" not in body, "packed single precision in the scoring loop"
    assert body.count("v_fmac_f32") + body.count("v_fma_f32") >= 4 * 14, "the single-precision tier is gone?"
    # no scratch traffic inside a loop: every line between a loop header comment and the branch back to it
    lines = body.splitlines()
    labels = {}
    for i, ln in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", ln)
        if mm:
            labels[mm.group(1)] = i
    in_loop = [False] * len(lines)
    for i, ln in enumerate(lines):
        mm = re.search(r"\bs_c?branch\S*\s+(\.LBB\d+_\d+)", ln)
        if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:      # a backward branch closes a loop
            for k in range(labels[mm.group(1)], i + 1):
                in_loop[k] = True
    assert any(in_loop)
    hot = [ln for i, ln in enumerate(lines) if in_loop[i] and "scratch_" in ln]
    # (the double-precision fallback inside the hypothesis loop may reload a spilled ADDRESS: that path runs for one
    # wave-step in ~30; stores would mean live state is spilled around the hot code)
    assert not [ln for ln in hot if "scratch_store" in ln], hot[:4]
    assert len(hot) <= 8, hot
