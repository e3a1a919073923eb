#!/usr/bin/env python
"""ISA audit of k_gfstack_dma / k_gfstack_ws (beat_amd/csrc/gfshared.hip).

The kernel hides its slot/weight loads from hipcc (asm `global_load_ushort` / `global_load_dwordx2`
whose completion is awaited by a hand-placed `s_waitcnt vmcnt(0)` one step later).  hipcc treats
the destination registers as written when the asm statement ends, so nothing guarantees that it
does not copy, reuse or spill them while the load is in flight.  This script compiles the file
to assembly and checks, for every instance of the kernel, that no instruction between such a load
and the wait that covers it (following the loop back-edge) names the destination registers.

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
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_ZN7beatamd1[23]k_gfstack_(?:dma|ws)\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            s = ln.strip()
            if s and not s.startswith(";") and not s.startswith("."):
                body.append(s.split(";")[0].strip())
            elif re.match(r"^\.LBB\d+_\d+:", s):
                body.append(s.split(";")[0].strip())
            if s.startswith("s_endpgm"):
                yield name, body
                name = None


def audit(body):
    labels = {ln[:-1]: i for i, ln in enumerate(body) if ln.endswith(":")}
    problems, nchecked = [], 0
    for i, ln in enumerate(body):
        if not (ln.startswith("global_load_ushort") or ln.startswith("global_load_dwordx2")):
            continue
        dest = vregs(ln.split(",")[0])
        nchecked += 1
        j, jumped, steps = i + 1, False, 0
        while j < len(body) and steps < 20000:
            cur = body[j]
            steps += 1
            if cur.startswith("s_waitcnt vmcnt(0)"):
                break
            if not cur.endswith(":"):
                # the partner load of the same statement group may use the register as address
                # BEFORE overwriting it; any other mention is a hazard
                ops = cur.split(None, 1)[1] if " " in cur else ""
                if vregs(ops) & dest and not cur.startswith("global_load_"):
                    problems.append((i, ln, j, cur))
                m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", cur)
                if m and not jumped and labels.get(m.group(1), 1 << 30) < j and \
                        labels[m.group(1)] < i:
                    # the loop back-edge: continue at the loop header (once)
                    nxt = j + 1
                    # keep walking the fall-through first only if it is the loop exit: the wait
                    # of interest is at the header
                    j, jumped = labels[m.group(1)], True
                    continue
            j += 1
    return nchecked, problems


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
    for name, body in functions(text.splitlines()):
        n, problems = audit(body)
        nfun += 1
        total += n
        for (i, ln, j, cur) in problems:
            bad += 1
            print("%s: load `%s` (line %d) in flight, touched by `%s` (line %d)" % (name, ln, i, cur, j))
    print("audited %d kernel instances, %d hidden loads, %d hazards" % (nfun, total, bad))
    return 1 if (bad or nfun == 0 or total == 0) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else None))
