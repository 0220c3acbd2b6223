"""Independent check of the MFMA wait states in the matcher's gfx950 ISA (VERDICT / ADVICE round 3, item "MFMA wrong key").

An XDL MFMA of P passes writes its result P passes after issue; gfx950 has no interlock for a VALU / LDS / VMEM instruction
(or the A / B operand of another MFMA) that touches those registers earlier: software owes P + 4 wait states (8-pass
`v_mfma_i32_32x32x32_i8`: 12, i.e. `s_nop 11`; /opt/skills/guides/cdna_hip_programming.md section 5.7 item 2, LLVM
GCNHazardRecognizer's GFX940_XDL_N_PassWriteVgprVALU*WaitStates + 1 on gfx950).  The compiler's hazard recogniser places
them -- unless the read is hidden from it (inline asm) or it is wrong.  This script does NOT trust it: it rebuilds the
control-flow graph of every kernel that holds a `v_mfma`, walks every path forward from every MFMA and reports

  * the smallest number of wait states between an MFMA and the first instruction that touches its destination registers
    (any operand position; a following MFMA whose SrcC AND destination are exactly the same registers is the
    accumulate chain the hardware does interlock and is allowed),
  * `v_accvgpr_*` moves (the accumulators left the VGPR form: csrc/Makefile's -mllvm -amdgpu-mfma-vgpr-form is gone or
    the register allocator fell back), and
  * conditional branches BETWEEN the MFMAs of one accumulation chain (the round-3 form that returned wrong keys under
    load had one around every MFMA).

    python scripts/check_mfma_hazards.py file.s [...]          (assembly from hipcc -S --cuda-device-only)
    python scripts/check_mfma_hazards.py --hip csrc/match.hip   (compiles with the Makefile's flags first)
Exit status 1 when a kernel violates a rule (tests/test_mfma_isa.py runs it on the matcher in the CPU suite).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {"32x32x32_i8": 8, "16x16x64_i8": 4, "32x32x16_i8": 8, "16x16x32_i8": 4}   # the integer XDL forms this library may use
MAKE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def regs_of(tok):
    """'v[82:85]' -> {('v',82),...}; 'a3' -> {('a',3)}; anything else -> empty."""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


class Inst(object):
    __slots__ = ("op", "ops", "text", "ws", "target", "cond")

    def __init__(self, text):
        self.text = text.strip()
        parts = self.text.split(None, 1)
        self.op = parts[0]
        rest = parts[1].split(";")[0] if len(parts) > 1 else ""
        self.ops = [t.strip() for t in rest.split(",") if t.strip()]
        self.ws = int(self.ops[0], 0) + 1 if self.op == "s_nop" else 1
        self.target, self.cond = None, False
        if self.op.startswith("s_cbranch"):
            self.target, self.cond = self.ops[0], True
        elif self.op == "s_branch":
            self.target = self.ops[0]


def kernels_in(asm_text):
    """{symbol: [lines]} of the functions that end in s_endpgm."""
    lines = asm_text.split("\n")
    out = {}
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l):
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            body = lines[i + 1:j]
            if any("s_endpgm" in b for b in body):
                out[l.split(":")[0]] = body
    return out


def analyse(body):
    """-> dict(mfma, accvgpr, branches_inside_chains, min_wait, violations[list of str])."""
    # flat instruction list + label -> index
    insts, labels = [], {}
    for l in body:
        s = l.split(";")[0].rstrip()
        if not s.strip():
            continue
        m = re.match(r"^(\.L\w+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if s.startswith("\t") or s.startswith(" "):
            t = s.strip()
            if t.startswith("."):
                continue
            insts.append(Inst(t))
    n = len(insts)

    def succ(i):
        ins = insts[i]
        if ins.op == "s_endpgm":
            return []
        if ins.target is not None:
            tgt = labels.get(ins.target)
            out = [tgt] if tgt is not None else []
            if ins.cond and i + 1 < n:
                out.append(i + 1)
            return out
        return [i + 1] if i + 1 < n else []

    res = {"mfma": 0, "accvgpr": 0, "branches_inside_chains": 0, "min_wait": None, "violations": []}
    mfmas = [i for i, ins in enumerate(insts) if ins.op.startswith("v_mfma")]
    res["mfma"] = len(mfmas)
    res["accvgpr"] = sum(1 for ins in insts if ins.op.startswith("v_accvgpr"))
    for i in mfmas:
        ins = insts[i]
        shape = ins.op.split("v_mfma_i32_")[-1] if "v_mfma_i32_" in ins.op else None
        need = PASSES.get(shape, 16) + 4
        dst = regs_of(ins.ops[0])
        # every path forward, until `need` wait states have gone by
        best = {}
        stack = [(s, 0) for s in succ(i)]
        while stack:
            j, dist = stack.pop()
            if dist >= need or j >= n:
                continue
            if best.get(j, 1 << 30) <= dist:
                continue
            best[j] = dist
            nxt = insts[j]
            touched = set()
            for tok in nxt.ops:
                touched |= regs_of(tok)
            if touched & dst:
                chain = (nxt.op.startswith("v_mfma") and len(nxt.ops) >= 4 and regs_of(nxt.ops[0]) == dst and regs_of(nxt.ops[3]) == dst
                         and not ((regs_of(nxt.ops[1]) | regs_of(nxt.ops[2])) & dst))
                if not chain:
                    res["violations"].append("%d wait states (need %d) between `%s` and `%s`" % (dist, need, ins.text, nxt.text))
                    res["min_wait"] = dist if res["min_wait"] is None else min(res["min_wait"], dist)
                    continue
                continue   # the chain's next link takes over (it is analysed on its own)
            for s in succ(j):
                stack.append((s, dist + nxt.ws))
        # the first touch beyond `need` is fine; record the smallest legal distance as well (informational)
    # conditional branches between two MFMAs that accumulate into the same registers
    by_dst = {}
    for i in mfmas:
        by_dst.setdefault(insts[i].ops[0], []).append(i)
    for dst_tok, idx in by_dst.items():
        for a, b in zip(idx, idx[1:]):
            if len(insts[b].ops) >= 4 and insts[b].ops[3] == dst_tok:   # b accumulates onto a's result
                res["branches_inside_chains"] += sum(1 for k in range(a + 1, b) if insts[k].cond)
    return res


def check_text(asm_text, only=None):
    report, bad = {}, False
    for sym, body in kernels_in(asm_text).items():
        if only and only not in sym:
            continue
        r = analyse(body)
        if r["mfma"] == 0:
            continue
        report[sym] = r
        if r["violations"] or r["accvgpr"] or r["branches_inside_chains"]:
            bad = True
    return report, bad


def compile_hip(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + MAKE_FLAGS + list(extra) +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(os.path.abspath(src)),
                               "--cuda-device-only", "-S", "-o", asm, src], stderr=subprocess.DEVNULL)
        return open(asm).read()


def main(argv):
    texts = []
    args = list(argv)
    while args:
        a = args.pop(0)
        if a == "--hip":
            texts.append((a, compile_hip(args.pop(0))))
        else:
            texts.append((a, open(a).read()))
    bad_any = False
    for name, text in texts:
        report, bad = check_text(text)
        bad_any |= bad
        for sym, r in report.items():
            print("%s: %d MFMA, %d v_accvgpr moves, %d conditional branches inside accumulation chains, %d early touches%s"
                  % (sym[:90], r["mfma"], r["accvgpr"], r["branches_inside_chains"], len(r["violations"]),
                     "" if not r["violations"] else " (min %d wait states)" % r["min_wait"]))
            for v in r["violations"][:6]:
                print("    " + v)
    return 1 if bad_any else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
