"""Static VALU instruction mix of the kernels in one .hip file (gfx950 assembly from `hipcc -S`), and the issue-bound
cycles per wave-instruction that follows from it with the PER-OPCODE rates measured by scripts/valu_rate.hip on the
MI355X (gpurun_out/valu_rate_r2.txt, round 2): the fast class (and / or / xor / not / add / sub / logical and
arithmetic right shifts / mov / v_bitop3 / 16-bit min, max, sub / f32 add, mul, fma) issues every ~2.7 SIMD-cycles at
the 2.4 GHz the figures are normalised to, everything else (bcnt, bfi, xnor, LEFT shifts, 32-bit min / max, compares,
SDWA / DPP forms, 64-bit shifts, f64, multiplies, ...) every ~4.4.  The encoding (e32 / e64) does not decide the class:
v_bitop3 and v_fma_f32 are VOP3 and fast, v_min_u32 and v_lshlrev_b32 are VOP2 and slow.
For the median kernel the 11-times unrolled row loop IS the kernel, so the static mix is the dynamic one.

    python scripts/isa_mix.py vo_single_camera_sos_amd/csrc/image.hip [-o profiles/roundN/<tag>_isa_mix.json]
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CYC_E32, CYC_OTHER = 2.7, 4.4   # class means; per-opcode values below where measured
FAST = {"v_and_b32": 2.79, "v_xor_b32": 2.84, "v_or_b32": 2.63, "v_not_b32": 2.58, "v_add_u32": 2.79, "v_sub_u32": 2.76,
        "v_subrev_u32": 2.76, "v_lshrrev_b32": 2.51, "v_ashrrev_i32": 2.60, "v_mov_b32": 2.57, "v_bitop3_b32": 2.67,
        "v_add_f32": 2.79, "v_mul_f32": 2.55, "v_fma_f32": 2.81, "v_min_u16": 2.51, "v_max_u16": 2.51, "v_sub_u16": 2.77,
        "v_max_f16": 2.69}


def op_cycles(op, line):
    base = re.sub(r"_(e32|e64)$", "", op)
    if "sdwa" in op or "dpp" in op or " row_" in line or " wave_" in line:
        return CYC_OTHER
    return FAST.get(base, CYC_OTHER)


def demangled_label(sym):
    m = re.search(r"N_1(\d+)", sym)
    if not m:
        return sym
    n = int(m.group(1))
    i = m.end()
    name = sym[i:i + n]
    rest = sym[i + n:]
    t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
    if t:
        name += "<" + ", ".join(re.findall(r"L[ib](\d+)E", t.group(1))) + ">"
    return name


def main():
    src = sys.argv[1]
    out = sys.argv[sys.argv.index("-o") + 1] if "-o" in sys.argv else None
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", asm, src],
                              stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    res = {}
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for i, sym in starts:
        j = i
        while "s_endpgm" not in lines[j]:
            j += 1
        c = collections.Counter()
        total = 0.0
        for l in lines[i:j]:
            m = re.match(r"^\s+(v_\w+)", l)
            if m:
                cy = op_cycles(m.group(1), l)
                c["fast" if cy < 3.5 else "slow"] += 1
                total += cy
        n = c["fast"] + c["slow"]
        if n == 0:
            continue
        res[demangled_label(sym)] = {"valu_static": n, "fast": c["fast"], "slow": c["slow"],
                                     "issue_cycles_per_valu_inst": round(total / n, 3)}
    doc = {"source": os.path.relpath(os.path.abspath(src), ROOT), "cycles_fast_class": CYC_E32, "cycles_slow_class": CYC_OTHER,
           "per_opcode_cycles": FAST, "rates_from": "scripts/valu_rate.hip on MI355X (profiles/round2/valu_rate.txt)",
           "kernels": res}
    txt = json.dumps(doc, indent=1)
    if out:
        with open(out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
