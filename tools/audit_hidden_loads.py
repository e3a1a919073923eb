#!/usr/bin/env python
"""ISA audit of k_gfstack_dma / k_gfstack_ws (beat_amd/csrc/gfshared.hip).

The kernels hide their slot/weight loads from hipcc (asm `global_load_ushort` / `global_load_dwordx2`
whose completion is awaited by a hand-placed `s_waitcnt vmcnt` one step later).  hipcc treats
the destination registers as written when the asm statement ends, so nothing guarantees that it
does not copy, reuse or spill them while the load is in flight.  This script compiles the file
to assembly and walks, for every such load of every instance, the control-flow graph forward --
fall-through, both arms of every conditional branch, the loop back-edge AND the ways out of the
loop -- until a vmcnt wait: no instruction on the way may name the destination registers.
(Round 2: the first version followed the back-edge only; the loads of the step after the last were
in flight behind the loop of k_gfstack_dma while its epilogue reused their registers.)

usage: audit_hidden_loads.py [path/to/gfshared.s]   (without an argument it runs hipcc -S)
exit status 0 = clean."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def functions(lines):
    """-> (name, body, in_asm): instructions and labels of every kernel instance; in_asm[i] tells
    whether body[i] comes from an inline-asm statement (between ;;#ASMSTART and ;;#ASMEND)"""
    name, body, in_asm, asm = None, [], [], False
    for ln in lines:
        m = re.match(r"^(_ZN7beatamd1[23]k_gfstack_(?:dma|ws)\w+):", ln)
        if m:
            name, body, in_asm, asm = m.group(1), [], [], False
            continue
        if name is not None:
            s = ln.strip()
            if s.startswith(";;#ASMSTART"):
                asm = True
            elif s.startswith(";;#ASMEND"):
                asm = False
            elif s and not s.startswith(";") and not s.startswith("."):
                body.append(s.split(";")[0].strip())
                in_asm.append(asm)
            elif re.match(r"^\.LBB\d+_\d+:", s):
                body.append(s.split(";")[0].strip())
                in_asm.append(False)
            if s.startswith("s_endpgm"):
                yield name, body, in_asm
                name = None


def audit(body, in_asm):
    """every hidden load (a global load written in an asm statement): walk the control-flow graph
    forward from it -- fall-through, both arms of conditional branches, loop back-edges and loop
    exits alike -- until a vmcnt wait; any instruction on the way that names the destination
    registers (other than a hidden load using them as its address) is a hazard, and so is
    reaching the end of a path that re-issues the same load (no wait in a whole loop round)"""
    labels = {ln[:-1]: i for i, ln in enumerate(body) if ln.endswith(":")}
    problems, nchecked = [], 0
    for i, ln in enumerate(body):
        if not in_asm[i] or not (ln.startswith("global_load_ushort") or ln.startswith("global_load_dwordx2")):
            continue
        dest = vregs(ln.split(",")[0])
        nchecked += 1
        seen, todo = set(), [i + 1]
        while todo:
            j = todo.pop()
            while j < len(body) and j not in seen:
                seen.add(j)
                cur = body[j]
                if re.match(r"s_waitcnt vmcnt\(\d+\)", cur) or cur.startswith("s_endpgm"):
                    # (the three-buffer variants wait with vmcnt(k), k = row requests issued AFTER
                    # the slot/weight loads: the loads themselves are covered by every arm)
                    break
                if not cur.endswith(":"):
                    ops = cur.split(None, 1)[1] if " " in cur else ""
                    if j == i:
                        problems.append((i, ln, j, "the same load again: a loop round without a wait"))
                        break
                    # the partner load of the same statement group may use the register as its
                    # address BEFORE overwriting it; any other mention is a hazard
                    if vregs(ops) & dest and not (in_asm[j] and cur.startswith("global_load_")):
                        problems.append((i, ln, j, cur))
                        break
                    m = re.match(r"s_(c?)branch\w*\s+(\.LBB\d+_\d+)", cur)
                    if m and m.group(2) in labels:
                        todo.append(labels[m.group(2)])
                        if not m.group(1):
                            break            # unconditional: no fall-through
                j += 1
    return nchecked, problems


def audit_order(body):
    """Three-buffer variants (any `s_waitcnt vmcnt(N)`, N > 0, in the kernel): the step-top wait
    leaves the N youngest requests in flight, so inside a step the slot/weight loads must be
    issued BEFORE the row requests (global_load_lds).  -> list of offending barrier positions."""
    if not any(re.match(r"s_waitcnt vmcnt\([1-9]\d*\)", ln) for ln in body):
        return []
    bad = []
    bars = [i for i, ln in enumerate(body) if ln.startswith("s_barrier")]
    for a, b in zip(bars, bars[1:] + [len(body)]):
        seg = body[a:b]
        tab = [i for i, ln in enumerate(seg) if ln.startswith("global_load_ushort")]
        dma = [i for i, ln in enumerate(seg) if ln.startswith("global_load_lds")]
        if tab and dma and min(dma) < min(tab):
            bad.append(a)
    return bad


def main(path=None):
    if path:
        text = open(path).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "gfshared.s")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC",
                                   "-S", "--cuda-device-only", os.path.join(ROOT, "beat_amd/csrc/gfshared.hip"),
                                   "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read()
    total, bad, nfun = 0, 0, 0
    for name, body, in_asm in functions(text.splitlines()):
        n, problems = audit(body, in_asm)
        nfun += 1
        total += n
        for (i, ln, j, cur) in sorted(set(problems)):
            bad += 1
            print("%s: load `%s` (line %d) in flight, touched by `%s` (line %d)" % (name, ln, i, cur, j))
        for a in audit_order(body):
            bad += 1
            print("%s: row requests issued before the slot/weight loads after the barrier at line %d" % (name, a))
    print("audited %d kernel instances, %d hidden loads, %d hazards" % (nfun, total, bad))
    return 1 if (bad or nfun == 0 or total == 0) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else None))
