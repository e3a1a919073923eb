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


def main_loop(body, labels):
    """(index of the back-edge branch, index of the header label) of the step loop: the backward
    branch whose span holds the most LDS reads (the gather).  Other backward branches (short DMA
    loops, out-of-line blocks) are not loop back-edges of interest."""
    best = None
    for j, cur in enumerate(body):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", cur)
        if not m or labels.get(m.group(1), 1 << 30) >= j:
            continue
        h = labels[m.group(1)]
        nds = sum(1 for ln in body[h:j] if ln.startswith("ds_read_b"))
        if nds >= 32 and (best is None or nds > best[2]):
            best = (j, h, nds)
    return best


def audit(body):
    labels = {ln[:-1]: i for i, ln in enumerate(body) if ln.endswith(":")}
    loop = main_loop(body, labels)
    problems, nchecked = [], 0
    for i, ln in enumerate(body):
        if not (ln.startswith("global_load_ushort") or ln.startswith("global_load_dwordx2")):
            continue
        if loop is None or i > loop[0]:
            continue   # epilogue loads are ordinary, waited loads
        dest = vregs(ln.split(",")[0])
        nchecked += 1
        j, jumped, steps = i + 1, False, 0
        while j < len(body) and steps < 20000:
            cur = body[j]
            steps += 1
            if re.match(r"s_waitcnt vmcnt\(\d+\)", cur):
                # (the three-buffer variants wait with vmcnt(k), k = row requests issued AFTER the
                # slot/weight loads: the loads themselves are covered by every arm of that switch)
                break
            if not cur.endswith(":"):
                # the partner load of the same statement group may use the register as address
                # BEFORE overwriting it; any other mention is a hazard
                ops = cur.split(None, 1)[1] if " " in cur else ""
                if vregs(ops) & dest and not cur.startswith("global_load_"):
                    problems.append((i, ln, j, cur))
                if j == loop[0] and not jumped:
                    j, jumped = loop[1], True   # the step loop's back-edge: on to the loop header
                    continue
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
    for name, body in functions(text.splitlines()):
        n, problems = audit(body)
        nfun += 1
        total += n
        for (i, ln, j, cur) in problems:
            bad += 1
            print("%s: load `%s` (line %d) in flight, touched by `%s` (line %d)" % (name, ln, i, cur, j))
        for a in audit_order(body):
            bad += 1
            print("%s: row requests issued before the slot/weight loads after the barrier at line %d" % (name, a))
    print("audited %d kernel instances, %d hidden loads, %d hazards" % (nfun, total, bad))
    return 1 if (bad or nfun == 0 or total == 0) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else None))
