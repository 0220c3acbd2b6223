"""Static VALU instruction mix of the kernels in .hip files (gfx950 assembly from `hipcc -S`), and the issue-bound
cost per wave-instruction that follows from it with the PER-OPCODE rates measured on the MI355X.

Two classes (scripts/valu_rate.hip, ~70 opcodes, round 2; scripts/valu_clock.hip, round 3, in REAL shader cycles with the
in-kernel clock s_memtime against s_memrealtime and the waves per SIMD pinned): the fast class (and / or / xor / not /
add / sub / logical and arithmetic right shifts / mov / v_bitop3 / 16-bit min, max, sub / f32 add, mul, fma) issues one
wave64 instruction per 2.50 shader cycles per SIMD at 2-4 waves per SIMD (2.25 at 8) while the clock sits at 2.2-2.35 GHz;
everything else (bcnt, bfi, xnor, LEFT shifts, 32-bit min / max, compares, SDWA / DPP forms, 64-bit shifts, f64,
multiplies, ...) one per 4.2-4.3 cycles at 2.37-2.39 GHz.  The encoding (e32 / e64) does not decide the class: v_bitop3 and
v_fma_f32 are VOP3 and fast, v_min_u32 and v_lshlrev_b32 are VOP2 and slow.  In TIME that is 1.11 ns and 1.79 ns per
wave-instruction per SIMD -- the same figures round 2 quoted as "2.7 / 4.4 cycles at an assumed 2.4 GHz" (the per-opcode
table below keeps that normalisation: cycles at 2.4 GHz = ns x 2.4, so that bench.py's bounds stay comparable).
For the median kernel the 11-times unrolled row loop IS the kernel, so the static mix is the dynamic one; for kernels with
data-dependent loops the static mix is only the best available price of an instruction.

    python scripts/isa_mix.py vo_single_camera_sos_amd/csrc/image.hip [more.hip ...] [-o profiles/roundN/<tag>_isa_mix.json]
    python scripts/isa_mix.py --all -o profiles/roundN/<tag>_isa_mix.json        (every .hip of the library)
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
        "v_add_f32": 2.79, "v_mul_f32": 2.55, "v_fma_f32": 2.81, "v_fmac_f32": 2.64, "v_sub_f32": 2.63, "v_subrev_f32": 2.63, "v_min_u16": 2.51, "v_max_u16": 2.51, "v_sub_u16": 2.77,
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


def kernels_of(src):
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
        while j < len(lines) and "s_endpgm" not in lines[j]:
            j += 1
        if j >= len(lines):
            continue   # (a host-side symbol, not a kernel)
        c = collections.Counter()
        total = 0.0
        for l in lines[i:j]:
            m = re.match(r"^\s+(v_\w+)", l)
            if m and m.group(1).startswith("v_mfma"):
                c["mfma"] += 1   # issues on the matrix pipe, beside the VALU
                continue
            if m:
                cy = op_cycles(m.group(1), l)
                c["fast" if cy < 3.5 else "slow"] += 1
                total += cy
        n = c["fast"] + c["slow"]
        if n == 0:
            continue
        res[demangled_label(sym)] = {"valu_static": n, "fast": c["fast"], "slow": c["slow"], "mfma_static": c["mfma"],
                                     "issue_cycles_per_valu_inst": round(total / n, 3)}
    return res


def main():
    out = sys.argv[sys.argv.index("-o") + 1] if "-o" in sys.argv else None
    if "--all" in sys.argv:
        import glob
        srcs = sorted(glob.glob(os.path.join(ROOT, "vo_single_camera_sos_amd", "csrc", "*.hip")))
    else:
        srcs = [a for i, a in enumerate(sys.argv[1:], 1) if a.endswith(".hip") and sys.argv[i - 1] != "-o"]
    res = {}
    for src in srcs:
        res.update(kernels_of(src))
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    doc = {"source": [os.path.relpath(os.path.abspath(x), ROOT) for x in srcs], "commit": commit or None,
           "cycles_fast_class": CYC_E32, "cycles_slow_class": CYC_OTHER, "normalised_to_GHz": 2.4,
           "real_cycles": {"fast": 2.50, "slow": 4.27, "shader_clock_GHz": "2.2-2.4 under load", "source": "profiles/round3/valu_clock.txt"},
           "per_opcode_cycles": FAST, "rates_from": "scripts/valu_rate.hip (round 2) and scripts/valu_clock.hip (round 3) on MI355X",
           "kernels": res}
    txt = json.dumps(doc, indent=1)
    if out:
        with open(out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
