"""Static VALU instruction mix of the kernels in one .hip file (gfx950 assembly from `hipcc -S`), and the issue-bound
cycles per wave-instruction that follows from it with the rates measured by scripts/valu_rate.hip on the MI355X:
2.8 SIMD-cycles for a plain VOP2 / VOP1 encoding (`_e32`), 4.4 for everything else (VOP3, SDWA, DPP, 64-bit shifts).
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
CYC_E32, CYC_OTHER = 2.8, 4.4


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
        for l in lines[i:j]:
            m = re.match(r"^\s+(v_\w+)", l)
            if m:
                op = m.group(1)
                c["e32" if op.endswith("_e32") and "dpp" not in l and "sdwa" not in l else "other"] += 1
        n = c["e32"] + c["other"]
        if n == 0:
            continue
        cyc = (CYC_E32 * c["e32"] + CYC_OTHER * c["other"]) / n
        res[demangled_label(sym)] = {"valu_static": n, "e32": c["e32"], "other": c["other"],
                                     "issue_cycles_per_valu_inst": round(cyc, 3)}
    doc = {"source": os.path.relpath(os.path.abspath(src), ROOT), "cycles_e32": CYC_E32, "cycles_other": CYC_OTHER,
           "rates_from": "scripts/valu_rate.hip on MI355X", "kernels": res}
    txt = json.dumps(doc, indent=1)
    if out:
        with open(out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
