#!/usr/bin/env python
"""
gfcell_emu.py -- functional model of the k_gfstack_runs wavefront programs (gfruns_asm.inc) and a numpy twin of the
table builders (k_gc_order / k_gm_tables in beat_amd/csrc/gfcell.hip), row passes included.

The wavefront programs are register-allocated by hand and run from tables; their control flow (ring of three row
buffers, barrier count per wavefront, pointer arithmetic, the VGPR index register, scalar descriptor loads) is
checked here on the CPU -- tests/test_gfcell_program.py interprets the very instruction lists
tools/gen_gfruns_asm.py emits, for all 16 wavefronts of a workgroup, against a direct evaluation of the multilinear
stack (reference beat/ffi/base.py:663-704).  Timing, hazards and wait counts are NOT modelled (the GPU tests cover
the real thing); fma is evaluated as a*b+c on both sides.

Test infrastructure only: nothing in the product imports this module.
"""
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_gfruns_asm as gen  # noqa: E402

genruns = gen
U32 = np.uint32
MASK64 = (1 << 64) - 1
DEAD = 0xFFFFFFFF
NCH, NCONS, NLOAD = gen.NCHAIN, gen.NCONS, gen.NLOAD
CG, WAVES = NCONS * NCH, NCONS + NLOAD
LTABDW, LREQ = gen.LTABDW, gen.LREQ
TPITCH = 65 * 8
PARAM_BYTES = WAVES * 128
# row slots of an LDS buffer: three buffers behind the parameter blocks in 160 KB (an even number)
CAP = ((160 * 1024 - PARAM_BYTES) // (3 * 512)) // 2 * 2


# =============================================================================== memory
class Memory(object):
    def __init__(self):
        self.regions = []
        self.next = 0x100000000

    def alloc(self, nbytes, init=None):
        base = self.next
        buf = np.zeros(nbytes + 64, dtype=np.uint8)
        if init is not None:
            raw = np.ascontiguousarray(init).view(np.uint8).ravel()
            buf[:raw.size] = raw
        self.regions.append((base, buf))
        self.next += ((nbytes + 64 + 0xFFFF) // 0x10000 + 1) * 0x10000
        return base

    def _find(self, addr, n):
        for base, buf in self.regions:
            if base <= addr and addr + n <= base + buf.size:
                return buf, addr - base
        raise IndexError("global access outside any allocation: 0x%x (+%d)" % (addr, n))

    def read(self, addr, n):
        buf, o = self._find(addr, n)
        return buf[o:o + n]

    def write(self, addr, data):
        buf, o = self._find(addr, len(data))
        buf[o:o + len(data)] = data

    def array(self, base, dtype, count):
        buf, o = self._find(base, count * np.dtype(dtype).itemsize)
        return buf[o:o + count * np.dtype(dtype).itemsize].view(dtype)


# =============================================================================== tables (numpy twin)
def gf_tables_ml(st, du, st_min, st_dt, du_min, du_dt, D, S, T, P):
    """k_gf_tables, multilinear branch (gfstack.hip): row ids [C,T,P,4] and factors [C,T,P,4]"""
    C = st.shape[0]
    ds = (st - st_min) / st_dt                       # [C,T,P]
    dd = ((du - du_min) / du_dt)[:, None, :]         # [C,1,P]
    sc = np.ceil(ds).astype(np.int64)
    dc = np.ceil(dd).astype(np.int64) + np.zeros_like(sc)
    stf, rtf = sc - ds, dc - dd
    sf, df = sc - 1, dc - 1
    sc, sf = np.where(sc < 0, sc + S, sc), np.where(sf < 0, sf + S, sf)
    dc, df = np.where(dc < 0, dc + D, dc), np.where(df < 0, df + D, df)
    row0 = (np.arange(T)[None, :, None] * P + np.arange(P)[None, None, :]) * D
    ro = np.stack([(row0 + dc) * S + sc, (row0 + dc) * S + sf, (row0 + df) * S + sc, (row0 + df) * S + sf], -1)
    fa = np.stack([(1 - stf) * (1 - rtf), stf * (1.0 - rtf), (1 - stf) * rtf, stf * rtf], -1)
    return ro.astype(np.uint32), fa


def gc_cut(F0, F1, C, cg, ngroups):
    """numpy twin of k_gc_key0 / k_gc_cut / k_gc_members (gfcell.hip): the batch cut into ngroups groups of cg chain slots
    by recursive bisection -- a range of groups [lo, hi) with more than one group is split into its first (hi - lo) // 2
    groups and the rest along the key in which ITS chains spread wider (key 0 on a tie), the first part takes the
    (hi - lo) // 2 * cg chains that come first by (key, chain id) -> members [C] (group g = members[g*cg : (g+1)*cg])"""
    lo = np.zeros(C, dtype=np.int64)
    hi = np.full(C, ngroups, dtype=np.int64)
    allc = np.arange(C)
    widest = ngroups
    while widest > 1:
        nlo, nhi = lo.copy(), hi.copy()
        for a_, b_ in sorted(set(zip(lo.tolist(), hi.tolist()))):
            if b_ - a_ <= 1:
                continue
            ids = allc[(lo == a_) & (hi == b_)]
            f0, f1 = F0[ids], F1[ids]
            f = f1 if (f1.max() - f1.min() > f0.max() - f0.min()) else f0
            rank = np.argsort(np.lexsort((ids, f)))
            half = (b_ - a_) // 2
            left = rank < half * cg
            nlo[ids] = np.where(left, a_, a_ + half)
            nhi[ids] = np.where(left, a_ + half, b_)
        lo, hi = nlo, nhi
        widest -= widest // 2
    members = np.empty(C, dtype=np.int64)
    for g in range(ngroups):
        ids = allc[lo == g]
        members[g * cg:g * cg + ids.size] = ids
    return members


def gc_cut_chunked(F0, F1, C, cg):
    """the cut of a batch of any size (gfcell.hip launch_gc_cut): chunks of min(8192 // cg, 64) whole groups as the chains
    come, each cut by gc_cut on its own -> members [C]"""
    chunk = max(1, min(8192 // cg, 64)) * cg
    members = np.empty(C, dtype=np.int64)
    for base in range(0, C, chunk):
        n = min(chunk, C - base)
        members[base:base + n] = base + gc_cut(F0[base:base + n], F1[base:base + n], n, cg, (n + cg - 1) // cg)
    return members


def gc_order(rowoff, C, T, P, S, sort=True, keys=None, global_members=True):
    """numpy twin of the chain order (gfcell.hip): batches of several groups are cut into groups by gc_cut (C <= 8192, <= 64 groups),
    then inside a group (k_gc_order): bands of whole wavefronts by the first key -- as many as make a wavefront's chains a
    square piece of the group's extent in the two keys --, inside a band by the second; keys: optional (k0[C], k1[C]) of the
    caller (hypocentre strike / dip), else start-time indices at patches 0, P/2 (five bands)"""
    ngroups = (C + CG - 1) // CG
    order = np.full(ngroups * CG, DEAD, dtype=np.uint32)
    allc = np.arange(C)
    if keys is not None:
        F0, F1 = [np.where(np.abs(k) <= 1.79e308, k, 0.0) for k in keys]
    else:
        F0 = (rowoff[allc, 0, 0, 3] % S).astype(np.float64)
        F1 = (rowoff[allc, 0, P // 2, 3] % S).astype(np.float64)
    members = allc
    if sort and global_members and 1 < ngroups:
        members = gc_cut_chunked(F0, F1, C, CG)
    for g in range(ngroups):
        cs = members[g * CG:min(C, (g + 1) * CG)]
        if not sort:
            order[g * CG:g * CG + cs.size] = cs
            continue
        tid = np.arange(cs.size)
        f0, f1 = F0[cs], F1[cs]
        r0 = np.argsort(np.lexsort((tid, f0)))
        nw = (cs.size + NCH - 1) // NCH
        nb = 5
        if keys is not None:
            e0, e1 = f0.max() - f0.min(), f1.max() - f1.min()
            with np.errstate(all="ignore"):
                r = float(nw) * e0 / e1 if e1 > 0.0 else float(nw) * float(nw)
            if not (r <= float(nw) * float(nw)):       # (extents that overflow, inf / inf)
                r = float(nw) * float(nw)
            nb = max(1, min(nw, int(np.rint(np.sqrt(r)))))
        band = (r0 // NCH) * nb // nw
        r1 = np.argsort(np.lexsort((tid, f1, band)))
        order[g * CG + r1] = cs
    return order


PASS_ALLOC = 6     # passes per patch the device tables are sized for (k_gfstack_runs falls back to k_gfstack beyond)


def max_passes(D, S, cap=CAP):
    """upper bound of the row passes of a patch (sizes the twin's tables; the device sizes its tables for PASS_ALLOC passes
    and lets k_gfstack stand in beyond): one when the patch's D * (S + 1) dense slots fit a buffer; else at most one pass
    per ceil line plus the cuts of the lines that do not fit alone (a cut pass holds more than cap - 4 slots)"""
    dense = D * (S + 1)
    if dense <= cap:
        return 1
    return 2 * D + (4 * CG + max(1, cap - 4) - 1) // max(1, cap - 4) + 1


def req_bound(n, nlines, S):
    """gm_req_bound (gfcell.hip): the row requests n slots over nlines duration lines can take at most"""
    return n if S > 255 else n // 2 + 2 * nlines + 1


def passes_along_the_duration_axis(cells, D, S, cap):
    """the count phase of k_gm_tables (gfcell.hip), statement for statement: cells = ascending (B << 16 | A) keys of a
    (group, target, patch) -> {key: pass}.  Per duration line three bitsets over the start-time slots s' -- C: used as a
    ceil line, F: used as a floor line, X: ceil nodes of the line's cells --; a greedy over the ceil lines closes a pass
    when the next line's slots (or requests) do not fit; a line that does not fit alone is cut along the start-time axis."""
    S1, lim = S + 1, NLOAD * LREQ
    bC, bF, bX = [set() for _ in range(D)], [set() for _ in range(D)], [set() for _ in range(D)]
    for key in cells:
        sb, sa = key >> 16, key & 0xFFFF
        dc, sc, df = sb // S1, sb % S1, sa // S1
        bC[dc].update((sc, sc + 1))
        bF[df].update((sc, sc + 1))
        bX[dc].add(sc)

    def fl(d):
        return D - 1 if d == 0 else d - 1
    linepass, cellpass = {}, {}
    cur, n, nlines, first, opened = -1, 0, 0, -1, False
    for d in range(D):
        if not bX[d]:
            continue
        f = fl(d)
        wrap_c = opened and d == D - 1 and first == 0 and D > 1
        floor_in = opened and f != d and first <= f < d
        if f == d:
            add, lines_add = len(bC[d] | bF[d]), 1
        else:
            add = len(bC[d] | (bF[d] if wrap_c else set())) - (len(bF[d]) if wrap_c else 0)
            add += (len(bF[f] | bC[f]) - len(bC[f])) if floor_in else len(bF[f])
            lines_add = (0 if wrap_c else 1) + (0 if (floor_in and bC[f]) else 1)
        alone = add if f == d else len(bC[d]) + len(bF[f])
        if opened and (n + add > cap or req_bound(n + add, nlines + lines_add, S) > lim):
            opened = False
        if not opened:
            if alone > cap or req_bound(alone, 1 if f == d else 2, S) > lim:
                m, prev = 0, -2
                cur += 1
                for s_ in sorted(bX[d]):
                    addc = (1 if s_ == prev + 1 else 2) * (1 if f == d else 2)
                    if m and (m + addc > cap or req_bound(m + addc, 2, S) > lim):
                        cur += 1
                        m, prev = 0, -2
                    m += (1 if s_ == prev + 1 else 2) * (1 if f == d else 2)
                    prev = s_
                    cellpass[(d, s_)] = cur
                continue
            cur += 1
            opened, first, n, nlines = True, d, alone, (1 if f == d else 2)
        else:
            n += add
            nlines += lines_add
        linepass[d] = cur
    out = {}
    for key in cells:
        sb = key >> 16
        dc, sc = sb // S1, sb % S1
        out[key] = cellpass[(dc, sc)] if (dc, sc) in cellpass else linepass[dc]
    return out


def gm_tables(rowoff, fac, slips, order, C, T, P, D, S, nvar=1, cap=None):
    """numpy twin of k_gm_tables (gfcell.hip) -> dict(wtab, ltab, dtab, ucount [GT*P] rows the loaders move, npass [GT*P],
    nv [GT] vsteps, vmax, smax): per (group, target, patch) the chains' cells are cut into ROW PASSES (greedy over the
    cells in ascending (ceil-duration line, ceil start-time node) order, a pass holds at most `cap` row slots and
    2 * LREQ row requests); per pass and slip variable one STEP: the loaders' request lines (rows compactly in
    (line, node) order), the consumers' descriptor lines (chains of the pass in cell order at positions 0..n-1, pads
    into the scratch accumulator behind them) and weight records"""
    cap = CAP if cap is None else cap
    ngroups = order.size // CG
    GT, DS, S1 = ngroups * T, D * S, S + 1
    dense_n = D * S1
    slips = [slips] if nvar == 1 and not isinstance(slips, (list, tuple)) else list(slips)
    maxpass = max_passes(D, S, cap)
    vmax = P * maxpass
    smax = vmax * nvar
    wtab = np.zeros(GT * NCONS * (smax + 1) * gen.WSTRIDE + 8192, dtype=np.uint8)
    ltab = np.zeros(GT * (smax + 3) * NLOAD * LTABDW, dtype=np.uint32)
    dtab = np.zeros(GT * NCONS * (smax + 1) * gen.DLINE + 64, dtype=np.uint32)
    ucount = np.zeros(GT * P, dtype=np.uint32)
    npass_o = np.zeros(GT * P, dtype=np.uint32)
    nv = np.zeros(GT, dtype=np.uint32)

    def row_of(sl):
        d, s1 = divmod(sl, S1)
        return d * S + (s1 - 1 if s1 else S - 1)

    def slots_of(key):
        sb_, sa_ = key >> 16, key & 0xFFFF
        return sorted({sa_, sa_ + 1, sb_, sb_ + 1})

    for g in range(ngroups):
        ids = order[g * CG:(g + 1) * CG]
        live = ids != DEAD
        for t in range(T):
            gt = g * T + t
            v = 0
            for p in range(P):
                row0 = (t * P + p) * DS
                rel = np.zeros((CG, 4), dtype=np.int64)
                rel[live] = rowoff[ids[live], t, p].astype(np.int64) - row0
                dc, sc, df = rel[:, 0] // S, rel[:, 0] % S, rel[:, 2] // S
                sb, sa = dc * S1 + sc, df * S1 + sc
                key = np.where(live, (sb << 16) | sa, 0xFFFFFFFF)
                cells = sorted(set(int(x) for x in key[live]))
                # ---- passes: greedy over the cells in ascending order
                # one pass when every dense slot of a patch (and the requests they can take) fits a buffer -- what the
                # launcher decides from D and S alone --, else the count phase of k_gm_tables
                dense_n_ = D * S1
                if dense_n_ <= cap and (dense_n_ if S > 255 else dense_n_ // 2 + 2 * D + 1) <= NLOAD * LREQ:
                    pass_of = {kc: 0 for kc in cells}
                else:
                    pass_of = passes_along_the_duration_axis(cells, D, S, cap)
                npass = max(pass_of.values()) + 1 if pass_of else 1
                assert npass <= maxpass, (npass, maxpass)
                npass_o[gt * P + p] = npass
                cpass = np.array([pass_of.get(int(k_), -1) for k_ in key])
                moved = 0
                for k in range(npass):
                    inpass = live & (cpass == k)
                    dense = sorted({x for kc in cells if pass_of[kc] == k for x in slots_of(kc)})
                    cidx = {sl: i for i, sl in enumerate(dense)}
                    n = len(dense)
                    assert n <= cap
                    moved += n
                    reqs, i = [], 0
                    while i < n:
                        ra = row_of(dense[i])
                        if i + 1 < n:
                            rb = row_of(dense[i + 1])
                            if rb > ra and rb - ra < 256:
                                reqs.append(ra | ((rb - ra) << 16) | (i << 24))
                                i += 2
                                continue
                        reqs.append(ra | (i << 24))
                        i += 1
                    assert len(reqs) <= NLOAD * LREQ, (len(reqs), n)
                    for iv in range(nvar):
                        s = (v + k) * nvar + iv
                        for ll in range(NLOAD):
                            h = ltab[((gt * (smax + 3) + s) * NLOAD + ll) * LTABDW:][:LTABDW]
                            mine = reqs[ll::NLOAD]
                            h[0] = len(mine)
                            h[1] = p * DS
                            h[2:2 + len(mine)] = mine
                        ring = (s % 3) * cap
                        for w in range(NCONS):
                            lo = w * NCH
                            mem = [j for j in range(NCH) if inpass[lo + j]]
                            perm = sorted(mem, key=lambda jj: (int(key[lo + jj]), jj))
                            base_w = (gt * NCONS + w) * (smax + 1) + s
                            dtab[base_w * gen.DLINE + gen.D_NCH] = len(perm)     # chains of the wavefront in this step (early exit)
                            for r in range(NCH):
                                q = r % 4
                                rec = base_w * gen.WSTRIDE + (r // 8) * gen.PAIR + ((r // 4) % 2) * 8
                                dl = base_w * gen.DLINE + gen.ddword(r)
                                if r < len(perm):
                                    j = perm[r]
                                    kk_ = lo + j
                                    c = int(ids[kk_])
                                    wv = [fac[c, t, p, kk] * slips[iv][c, p] for kk in range(4)]
                                    nxt = 1 if (r + 1 < len(perm) and int(key[lo + perm[r + 1]]) != int(key[kk_])) else 0
                                    dtab[dl] = gen.D_BASE | j | (nxt << 31)
                                    dtab[dl + 1] = int(ring + cidx[int(sa[kk_])]) | (int(ring + cidx[int(sb[kk_])]) << 16)
                                else:
                                    wv = [0.0] * 4
                                    dtab[dl] = gen.D_BASE | gen.SCRATCH
                                    dtab[dl + 1] = int(ring) | (int(ring) << 16)
                                for kk in range(4):
                                    wtab[rec + (4 * q + kk) * 16:rec + (4 * q + kk) * 16 + 8].view(np.float64)[0] = wv[kk]
                ucount[gt * P + p] = moved
                v += npass
            nv[gt] = v
            # (the request lines behind the last step stay empty: the loaders read three steps ahead)
    return dict(wtab=wtab, ltab=ltab, dtab=dtab, ucount=ucount, npass=npass_o, nv=nv, vmax=vmax, smax=smax, cap=cap)


# =============================================================================== interpreter
class Barrier(Exception):
    pass


def _parse_program(kind, nth):
    lines = gen.consumer() if kind == "consumer" else gen.loader(nth)
    labels, prog = {}, []
    for ln in lines:
        m = re.match(r"^(\w+)_%=:$", ln)
        if m:
            labels[m.group(1)] = len(prog)
        else:
            prog.append(ln.replace("_%=", ""))
    return prog, labels


def _split_ops(rest):
    out, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Wave(object):
    def __init__(self, wg, wave_id, param_vgpr_value):
        self.wg = wg
        self.id = wave_id
        self.s = np.zeros(128, dtype=np.uint64)       # SGPR file (32-bit values)
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.v[0, :] = param_vgpr_value               # the "%0" input lives in v0
        self.exec = MASK64
        self.vcc = 0
        self.scc = 0
        self.m0 = 0
        self.idx_en = False
        self.pc = 0
        self.done = False
        self.nbarrier = 0
        self.ninstr = 0
        self.fma_count = 0
        self.pad_fma_count = 0
        # Loads are ASYNCHRONOUS: a destination register holds POISON from issue until the s_waitcnt that covers the load
        # (LDS and vector-memory operations return in order, scalar loads do not: with a scalar load in flight only
        # lgkmcnt(0) proves anything).  A program that uses a register before its wait computes NaN / indexes out of
        # range / selects a wild accumulator, and the tests that compare results fail; LDS-DMA rows reach LDS at the
        # loader's wait, not at issue.
        self.pend_lds = []      # [(commit function)] in issue order
        self.pend_sm = []
        self.pend_vm = []

    POISON_LO, POISON_HI, POISON_S = 0xDEADBEEF, 0x7FF8DEAD, 0xDEADBEEF

    def _wait(self, ln):
        mv = re.search(r"vmcnt\((\d+)\)", ln)
        ml = re.search(r"lgkmcnt\((\d+)\)", ln)
        if mv:
            n = int(mv.group(1))
            while len(self.pend_vm) > n:
                self.pend_vm.pop(0)()
        if ml:
            n = int(ml.group(1))
            if n == 0:
                for f in self.pend_lds + self.pend_sm:
                    f()
                del self.pend_lds[:], self.pend_sm[:]
            elif not self.pend_sm:
                while len(self.pend_lds) > n:
                    self.pend_lds.pop(0)()
            # (a counted wait with scalar loads in flight guarantees nothing)

    def _drain(self):
        for f in self.pend_vm + self.pend_lds + self.pend_sm:
            f()
        del self.pend_vm[:], self.pend_lds[:], self.pend_sm[:]

    def _load_v(self, queue, d, lanes, words):
        """vector destination registers d.. <- words [nreg][64] for `lanes`, at the wait; poison until then"""
        words = [np.array(w_, dtype=np.uint32) for w_ in words]
        for kk in range(len(words)):
            self.v[d + kk][lanes] = self.POISON_HI if (kk % 2 == 1) else self.POISON_LO

        def commit():
            for kk, w_ in enumerate(words):
                self.v[d + kk][lanes] = w_[lanes]
        queue.append(commit)

    # ---- operand access
    def sreg(self, name):
        if name == "vcc":
            return ("special", "vcc")
        if name == "exec":
            return ("special", "exec")
        if name == "m0":
            return ("special", "m0")
        m = re.match(r"^s\[(\d+):(\d+)\]$", name)
        if m:
            return ("s", int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1)
        m = re.match(r"^s(\d+)$", name)
        if m:
            return ("s", int(m.group(1)), 1)
        return None

    def get_s32(self, op):
        r = self.sreg(op)
        if r is None:
            return int(op, 0) & 0xFFFFFFFF
        if r[0] == "special":
            return getattr(self, r[1]) & 0xFFFFFFFF
        return int(self.s[r[1]]) & 0xFFFFFFFF

    def get_s64(self, op):
        r = self.sreg(op)
        if r is None:
            return int(op, 0) & MASK64
        if r[0] == "special":
            return getattr(self, r[1]) & MASK64
        return (int(self.s[r[1]]) & 0xFFFFFFFF) | ((int(self.s[r[1] + 1]) & 0xFFFFFFFF) << 32)

    def set_s32(self, op, val):
        r = self.sreg(op)
        val &= 0xFFFFFFFF
        if r[0] == "special":
            setattr(self, r[1], val if r[1] == "m0" else (getattr(self, r[1]) & ~0xFFFFFFFF) | val)
        else:
            self.s[r[1]] = val

    def set_s64(self, op, val):
        r = self.sreg(op)
        val &= MASK64
        if r[0] == "special":
            setattr(self, r[1], val)
        else:
            self.s[r[1]] = val & 0xFFFFFFFF
            self.s[r[1] + 1] = val >> 32

    @staticmethod
    def vreg(name):
        m = re.match(r"^-?v\[(\d+):(\d+)\]$", name)
        if m:
            return int(m.group(1))
        m = re.match(r"^-?v(\d+)$", name)
        if m:
            return int(m.group(1))
        return None

    def src32(self, op):
        """-> uint32[64]"""
        if op == "%0":
            return self.v[0]
        r = self.vreg(op)
        if r is not None:
            return self.v[r]
        return np.full(64, self.get_s32(op), dtype=np.uint32)

    def src_f64(self, op, rel=0):
        neg = op.startswith("-")
        r = self.vreg(op)
        if r is not None:
            r += rel
            x = (self.v[r].astype(np.uint64) | (self.v[r + 1].astype(np.uint64) << np.uint64(32))).view(np.float64)
        else:
            x = np.full(64, self.get_s64(op), dtype=np.uint64).view(np.float64)
        return -x if neg else x

    def lanes(self):
        return np.array([(self.exec >> i) & 1 for i in range(64)], dtype=bool)

    def wr32(self, r, val):
        m = self.lanes()
        x = (np.broadcast_to(np.asarray(val), (64,)).astype(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        self.v[r][m] = x[m]

    def wr_f64(self, r, x):
        m = self.lanes()
        u = np.asarray(x, dtype=np.float64).view(np.uint64)
        self.v[r][m] = (u & np.uint64(0xFFFFFFFF)).astype(np.uint32)[m]
        self.v[r + 1][m] = (u >> np.uint64(32)).astype(np.uint32)[m]

    # ---- execution
    def run(self):
        prog, labels = self.prog, self.labels
        lds, mem = self.wg.lds, self.wg.mem
        while self.pc < len(prog):
            ln = prog[self.pc]
            self.pc += 1
            self.ninstr += 1
            if self.ninstr > self.wg.max_instr:
                raise RuntimeError("wavefront %d: instruction budget exceeded (runaway loop?)" % self.id)
            parts = ln.split(None, 1)
            op = parts[0]
            rest = parts[1] if len(parts) > 1 else ""
            imm_off = 0
            mo = re.search(r"\soffset:(\w+)", rest)
            if mo:
                imm_off = int(mo.group(1), 0)
                rest = rest.replace(mo.group(0), "")
            ops = _split_ops(rest)
            # the VGPR index applies to vector ALU operands only (vector memory loads are issued under a non-zero index by
            # k_gfstack_runs and land in the registers named: bitwise-correct on the GPU); LDS instructions are kept
            # behind index 0 by the programs
            if self.idx_en and (self.m0 & 0xFF) and (op.startswith("v_") and op != "v_fmac_f64_dpp"
                                                      or op.startswith("ds_")):
                raise RuntimeError("%s executed with a non-zero VGPR index" % op)
            # ------------------------------------------------ scalar
            if op == "s_nop":
                continue
            if op == "s_waitcnt":
                self._wait(ln)
                continue
            if op == "s_barrier":
                self.nbarrier += 1
                yield "barrier"
                continue
            if op == "s_mov_b32":
                self.set_s32(ops[0], self.get_s32(ops[1]))
            elif op == "s_mov_b64":
                val = int(ops[1], 0) if self.sreg(ops[1]) is None else self.get_s64(ops[1])
                self.set_s64(ops[0], val)
            elif op in ("s_add_u32", "s_addc_u32", "s_sub_u32"):
                a, b = self.get_s32(ops[1]), self.get_s32(ops[2])
                if op == "s_add_u32":
                    r = a + b
                    self.scc = int(r > 0xFFFFFFFF)
                elif op == "s_addc_u32":
                    r = a + b + self.scc
                    self.scc = int(r > 0xFFFFFFFF)
                else:
                    r = a - b
                    self.scc = int(b > a)
                self.set_s32(ops[0], r)
            elif op == "s_mul_i32":
                self.set_s32(ops[0], self.get_s32(ops[1]) * self.get_s32(ops[2]))
            elif op == "s_mul_hi_u32":
                self.set_s32(ops[0], (self.get_s32(ops[1]) * self.get_s32(ops[2])) >> 32)
            elif op in ("s_and_b32", "s_or_b32", "s_lshr_b32", "s_lshl_b32"):
                a, b = self.get_s32(ops[1]), self.get_s32(ops[2])
                r = {"s_and_b32": a & b, "s_or_b32": a | b, "s_lshr_b32": a >> (b & 31),
                     "s_lshl_b32": (a << (b & 31))}[op] & 0xFFFFFFFF
                self.set_s32(ops[0], r)
                self.scc = int(r != 0)
            elif op == "s_and_b64":
                r = self.get_s64(ops[1]) & self.get_s64(ops[2])
                self.set_s64(ops[0], r)
                self.scc = int(r != 0)
            elif op.startswith("s_cmp_"):
                a, b = self.get_s32(ops[0]), self.get_s32(ops[1])
                self.scc = int({"eq": a == b, "le": a <= b, "lt": a < b, "ge": a >= b}[op.split("_")[2]])
            elif op == "s_bitcmp1_b32":
                self.scc = (self.get_s32(ops[0]) >> (self.get_s32(ops[1]) & 31)) & 1
            elif op == "s_bfe_u32":
                a_, f = self.get_s32(ops[1]), self.get_s32(ops[2])
                r = (a_ >> (f & 31)) & ((1 << ((f >> 16) & 0x7F)) - 1)
                self.set_s32(ops[0], r)
                self.scc = int(r != 0)
            elif op == "s_lshl_b64":
                r = (self.get_s64(ops[1]) << (self.get_s32(ops[2]) & 63)) & MASK64
                self.set_s64(ops[0], r)
                self.scc = int(r != 0)
            elif op == "s_or_b64":
                r = self.get_s64(ops[1]) | self.get_s64(ops[2])
                self.set_s64(ops[0], r)
                self.scc = int(r != 0)
            elif op == "s_cselect_b64":
                self.set_s64(ops[0], self.get_s64(ops[1]) if self.scc else self.get_s64(ops[2]))
            elif op == "s_bfm_b64":
                n, sh = self.get_s32(ops[1]) & 63, self.get_s32(ops[2]) & 63
                self.set_s64(ops[0], ((1 << n) - 1) << sh)
            elif op == "s_cselect_b32":
                self.set_s32(ops[0], self.get_s32(ops[1]) if self.scc else self.get_s32(ops[2]))
            elif op in ("s_cbranch_scc1", "s_cbranch_scc0", "s_branch"):
                if op == "s_branch" or self.scc == int(op.endswith("1")):
                    self.pc = labels[ops[0]]
            elif op == "s_set_gpr_idx_on":
                self.idx_en = True
                self.m0 = (self.m0 & ~0xF0FF) | (self.get_s32(ops[0]) & 0xFF) | ((int(ops[1], 0) & 0xF) << 12)
            elif op == "s_set_gpr_idx_off":
                self.idx_en = False
            elif op == "s_set_gpr_idx_idx":
                self.m0 = (self.m0 & ~0xFF) | (self.get_s32(ops[0]) & 0xFF)
            elif op.startswith("s_load_dword"):
                n = int(op[len("s_load_dwordx"):] or 1)
                r = self.sreg(ops[0])
                assert r[2] == n and r[1] % min(n, 4) == 0, ln
                base = self.sreg(ops[1])
                assert base[1] % 2 == 0, ln
                addr = self.get_s64(ops[1]) + int(ops[2], 0)
                vals = np.array(mem.read(addr, 4 * n).view(np.uint32), dtype=np.uint64)
                self.s[r[1]:r[1] + n] = self.POISON_S

                def commit_s(r0=r[1], n=n, vals=vals):
                    self.s[r0:r0 + n] = vals
                self.pend_sm.append(commit_s)
            # ------------------------------------------------ vector
            elif op == "v_mbcnt_lo_u32_b32":
                self.wr32(self.vreg(ops[0]), np.minimum(np.arange(64), 32) + self.src32(ops[2]))
            elif op == "v_mbcnt_hi_u32_b32":
                self.wr32(self.vreg(ops[0]), np.maximum(np.arange(64) - 32, 0) + self.src32(ops[2]))
            elif op == "v_lshlrev_b32":
                self.wr32(self.vreg(ops[0]), self.src32(ops[2]).astype(np.uint64) << np.uint64(self.get_s32(ops[1])))
            elif op == "v_lshrrev_b32":
                self.wr32(self.vreg(ops[0]), self.src32(ops[2]).astype(np.uint64) >> np.uint64(self.get_s32(ops[1])))
            elif op == "v_add_u32":
                self.wr32(self.vreg(ops[0]), self.src32(ops[1]).astype(np.uint64) + self.src32(ops[2]))
            elif op == "v_lshl_add_u32":
                self.wr32(self.vreg(ops[0]), (self.src32(ops[1]).astype(np.uint64) << np.uint64(self.get_s32(ops[2])))
                          + self.src32(ops[3]))
            elif op == "v_mad_u32_u16":
                # D = S0.u16 * S1.u16 + S2.u32; op_sel:[a,b,c,d]: high halves of src0 / src1
                sel = re.search(r"op_sel:\[(\d),(\d),(\d),(\d)\]", ln)
                o3 = re.sub(r"\s+op_sel:\[[\d,]+\]", "", ops[3]).strip()
                h0, h1 = (int(sel.group(1)), int(sel.group(2))) if sel else (0, 0)
                a_ = (self.src32(ops[1]).astype(np.uint64) >> (16 * h0)) & 0xFFFF
                b_ = (self.src32(ops[2]).astype(np.uint64) >> (16 * h1)) & 0xFFFF
                self.wr32(self.vreg(ops[0]), (a_ * b_ + self.src32(o3).astype(np.uint64)) & 0xFFFFFFFF)
            elif op == "v_mad_u32_u24":
                self.wr32(self.vreg(ops[0]), (self.src32(ops[1]).astype(np.uint64) & np.uint64(0xFFFFFF))
                          * (self.src32(ops[2]).astype(np.uint64) & np.uint64(0xFFFFFF)) + self.src32(ops[3]))
            elif op == "v_mul_u32_u24":
                self.wr32(self.vreg(ops[0]), (self.src32(ops[1]).astype(np.uint64) & np.uint64(0xFFFFFF))
                          * (self.src32(ops[2]).astype(np.uint64) & np.uint64(0xFFFFFF)))
            elif op == "v_mov_b32":
                self.wr32(self.vreg(ops[0]), self.src32(ops[1]).astype(np.uint64))
            elif op == "v_cndmask_b32":
                assert ops[3] == "vcc"
                sel = np.array([(self.vcc >> i) & 1 for i in range(64)], dtype=bool)
                self.wr32(self.vreg(ops[0]), np.where(sel, self.src32(ops[2]), self.src32(ops[1])).astype(np.uint64))
            elif op == "v_readlane_b32":
                lane_sel = self.get_s32(ops[2]) & 63
                self.set_s32(ops[0], int(self.v[self.vreg(ops[1])][lane_sel]))
            elif op == "v_and_b32":
                self.wr32(self.vreg(ops[0]), self.src32(ops[1]) & self.src32(ops[2]))
            elif op == "v_add_u32_dpp":
                k = int(re.search(r"row_newbcast:(\d+)", ln).group(1))
                ops = [o.split()[0] for o in ops]
                a_ = self.src32(ops[1])[(np.arange(64) // 16) * 16 + k]
                self.wr32(self.vreg(ops[0]), a_.astype(np.uint64) + self.src32(ops[2]))
            elif op == "v_fmac_f64_dpp":
                # D = dpp(S0) * S1 + D ; row_newbcast:k: lane k of the lane's own 16-lane row
                k = int(re.search(r"row_newbcast:(\d+)", ln).group(1))
                ops = [o.split()[0] for o in ops]
                assert self.idx_en and ((self.m0 >> 12) & 0xF) == 0x8, ln
                d = self.vreg(ops[0]) + (self.m0 & 0xFF)
                assert gen.ACC <= d <= gen.V_LAST - 1 and (d - gen.ACC) % 2 == 0, (ln, d)
                if d == gen.ACC + 2 * gen.SCRATCH:
                    self.pad_fma_count += 1
                a_ = self.src_f64(ops[1])[(np.arange(64) // 16) * 16 + k]
                b_ = self.src_f64(ops[2])
                c_ = (self.v[d].astype(np.uint64) | (self.v[d + 1].astype(np.uint64) << np.uint64(32))).view(np.float64)
                self.fma_count += 1
                with np.errstate(all="ignore"):
                    self.wr_f64(d, a_ * b_ + c_)
            elif op == "v_writelane_b32":
                self.v[self.vreg(ops[0])][int(ops[2])] = U32(self.get_s32(ops[1]))
            elif op == "v_fma_f64":
                rel = (self.m0 & 0xFF) if self.idx_en else 0
                mode = (self.m0 >> 12) & 0xF if self.idx_en else 0
                a = self.src_f64(ops[1], rel if mode & 1 else 0)
                b = self.src_f64(ops[2], rel if mode & 2 else 0)
                c = self.src_f64(ops[3], rel if mode & 4 else 0)
                d = self.vreg(ops[0]) + (rel if mode & 8 else 0)
                assert not self.idx_en
                with np.errstate(all="ignore"):
                    self.wr_f64(d, a * b + c)
            elif op == "v_add_f64":
                with np.errstate(all="ignore"):
                    self.wr_f64(self.vreg(ops[0]), self.src_f64(ops[1]) + self.src_f64(ops[2]))
            elif op == "v_mul_f64":
                with np.errstate(all="ignore"):
                    self.wr_f64(self.vreg(ops[0]), self.src_f64(ops[1]) * self.src_f64(ops[2]))
            elif op == "v_mad_u64_u32":
                r = self.vreg(ops[0])
                c = self.v[self.vreg(ops[4])].astype(np.uint64) | (self.v[self.vreg(ops[4]) + 1].astype(np.uint64) << np.uint64(32))
                prod = self.src32(ops[2]).astype(np.uint64) * self.src32(ops[3]).astype(np.uint64) + c
                m = self.lanes()
                self.v[r][m] = (prod & np.uint64(0xFFFFFFFF)).astype(np.uint32)[m]
                self.v[r + 1][m] = (prod >> np.uint64(32)).astype(np.uint32)[m]
            elif op == "v_cmp_ne_u32":
                ne = self.src32(ops[1]) != self.src32(ops[2])
                val = 0
                for i in range(64):
                    if ne[i] and (self.exec >> i) & 1:
                        val |= 1 << i
                self.set_s64(ops[0], val)
            # ------------------------------------------------ LDS
            elif op == "ds_write_b128":
                addr = self.v[self.vreg(ops[0])].astype(np.int64) + imm_off
                d = int(re.match(r"v\[(\d+):", ops[1]).group(1))
                for i in np.nonzero(self.lanes())[0]:
                    assert 0 <= addr[i] and addr[i] + 16 <= lds.size and addr[i] % 16 == 0, ("LDS write", ln, addr[i])
                    lds[addr[i]:addr[i] + 16].view(np.uint32)[:] = [self.v[d + kk][i] for kk in range(4)]
            elif op == "ds_read2_b64":
                # two 8-byte reads at addr + offset0 * 8 and addr + offset1 * 8 -> v[d:d+1], v[d+2:d+3]
                o0 = int(re.search(r"offset0:(\d+)", ln).group(1)) if "offset0:" in ln else 0
                o1 = int(re.search(r"offset1:(\d+)", ln).group(1)) if "offset1:" in ln else 0
                d = int(re.match(r"v\[(\d+):(\d+)\]", ops[0]).group(1))
                areg = re.sub(r"\s+offset[01]:\d+", "", ops[1]).strip()
                base_a = self.v[self.vreg(areg)].astype(np.int64)
                words = [np.zeros(64, dtype=np.uint32) for _ in range(4)]
                for half, o in enumerate((o0, o1)):
                    addr = base_a + 8 * o
                    for i in np.nonzero(self.lanes())[0]:
                        if not (0 <= addr[i] and addr[i] + 8 <= lds.size):
                            raise IndexError("LDS read out of range: %s lane %d addr %d" % (ln, i, addr[i]))
                        w = lds[addr[i]:addr[i] + 8].view(np.uint32)
                        words[2 * half][i], words[2 * half + 1][i] = w[0], w[1]
                self._load_v(self.pend_lds, d, self.lanes(), words)
            elif op in ("ds_read_b32", "ds_read_b64", "ds_write_b64"):
                off = imm_off
                m = self.lanes()
                if op == "ds_write_b64":
                    addr = self.v[self.vreg(ops[0])].astype(np.int64) + off
                    d = self.vreg(ops[1])
                    for i in np.nonzero(m)[0]:
                        assert 0 <= addr[i] and addr[i] + 8 <= lds.size, ("LDS write out of range", ln, addr[i])
                        lds[addr[i]:addr[i] + 8].view(np.uint32)[:] = (self.v[d][i], self.v[d + 1][i])
                else:
                    n = 4 if op == "ds_read_b32" else 8
                    addr = self.v[self.vreg(ops[1])].astype(np.int64) + off
                    d = self.vreg(ops[0])
                    words = [np.zeros(64, dtype=np.uint32) for _ in range(n // 4)]
                    for i in np.nonzero(m)[0]:
                        if not (0 <= addr[i] and addr[i] + n <= lds.size):
                            raise IndexError("LDS read out of range: %s lane %d addr %d" % (ln, i, addr[i]))
                        w = lds[addr[i]:addr[i] + n].view(np.uint32)
                        words[0][i] = w[0]
                        if n == 8:
                            words[1][i] = w[1]
                    self._load_v(self.pend_lds, d, m, words)
            # ------------------------------------------------ global
            elif op == "global_load_lds_dwordx4":
                base = self.get_s64(ops[1].replace(" nt", ""))
                voff = self.v[self.vreg(ops[0])].astype(np.int64)
                moves = []
                for i in np.nonzero(self.lanes())[0]:
                    dst = self.m0 + 16 * i
                    assert PARAM_BYTES <= dst and dst + 16 <= lds.size, ("LDS-DMA destination", ln, dst)
                    moves.append((dst, np.array(mem.read(base + int(voff[i]), 16))))
                    lds[dst:dst + 16] = 0xA5         # (not there yet)

                def commit_dma(moves=moves):
                    for dst, data in moves:
                        lds[dst:dst + 16] = data
                self.pend_vm.append(commit_dma)
                self.wg.dma_bytes += 16 * int(self.lanes().sum())
            elif op in ("global_load_dwordx2", "global_load_dword", "global_load_dwordx3", "global_load_dwordx4"):
                n = 16 if op.endswith("x4") else 12 if op.endswith("x3") else 8 if op.endswith("x2") else 4
                off = imm_off
                last = ops[2].split()[0]
                base = self.get_s64(last)
                voff = self.v[self.vreg(ops[1])].astype(np.int64)
                d = self.vreg(ops[0])
                words = [np.zeros(64, dtype=np.uint32) for _ in range(n // 4)]
                for i in np.nonzero(self.lanes())[0]:
                    w = mem.read(base + int(voff[i]) + off, n).view(np.uint32)
                    for kk in range(n // 4):
                        words[kk][i] = w[kk]
                self._load_v(self.pend_vm, d, self.lanes(), words)
            elif op == "global_store_dwordx2":
                d = self.vreg(ops[1])
                if ops[2] == "off":
                    a0 = self.vreg(ops[0])
                    addr = (self.v[a0].astype(np.uint64) | (self.v[a0 + 1].astype(np.uint64) << np.uint64(32))) + np.uint64(imm_off)
                else:
                    addr = np.uint64(self.get_s64(ops[2])) + self.v[self.vreg(ops[0])].astype(np.uint64) + np.uint64(imm_off)
                for i in np.nonzero(self.lanes())[0]:
                    mem.write(int(addr[i]), np.array([self.v[d][i], self.v[d + 1][i]], dtype=np.uint32).view(np.uint8))
            else:
                raise NotImplementedError(ln)
        self._drain()
        self.done = True


class Workgroup(object):
    """one (group g, target t, tile) workgroup of k_gfstack_runs: 14 consumers + 2 loaders"""

    def __init__(self, mem, nth, lds_bytes, params, max_instr=5000000):
        self.mem = mem
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.max_instr = max_instr
        self.dma_bytes = 0
        self.programs = {"consumer": _parse_program("consumer", nth), "loader": _parse_program("loader", nth)}
        self.waves = []
        for w in range(WAVES):
            self.lds[w * 128:(w + 1) * 128].view(np.uint32)[:] = params[w]
            wave = Wave(self, w, w * 128)
            wave.prog, wave.labels = self.programs["consumer" if w < NCONS else "loader"]
            self.waves.append(wave)

    def run(self):
        gens = [w.run() for w in self.waves]
        alive = list(range(WAVES))
        while alive:
            arrived = []
            for i in list(alive):
                try:
                    next(gens[i])
                    arrived.append(i)
                except StopIteration:
                    alive.remove(i)
            assert not arrived or len(arrived) == len(alive), "wavefronts disagree on the number of barriers"
        nb = {w.nbarrier for w in self.waves}
        assert len(nb) == 1, nb
        return self


def wave_params(w, g, t, tile, a):
    """the parameter block the C++ prologue of k_gfstack_runs writes (dict a: the kernel arguments + the tables of
    gm_tables)"""
    P = np.zeros(32, dtype=np.uint32)

    def put64(k, x):
        P[k], P[k + 1] = x & 0xFFFFFFFF, x >> 32
    gt = g * a["Ttab"] + (0 if a["Ttab"] == 1 else t)
    n0 = tile * 64
    N, T = a["N"], a["T"]
    smax = a["smax"]
    nsteps = int(a["nv"][gt]) * a.get("nvar", 1)
    if w < NCONS:
        put64(gen.P_WP, a["wtab"] + ((gt * NCONS + w) * (smax + 1)) * gen.WSTRIDE)
        P[gen.P_RB0] = PARAM_BYTES
        P[gen.P_NSTEP] = nsteps
        put64(gen.P_DP, a["dtab"] + ((gt * NCONS + w) * (smax + 1)) * gen.DLINE * 4)
        P[gen.P_MODE] = a["mode"]
        put64(gen.P_DATA, a["data"] + (t * N + n0) * 8)
        if a["mode"] == 3:       # bidiagonal misfit: edges, band rows, "the tile ends the trace" in the words of out / w / ctn
            put64(gen.P_OUT, a["edges"] + (t * a["ntile"] + tile) * 16)
            P[gen.P_CTN] = 1 if n0 + min(64, N - n0) == N else 0
            put64(gen.P_W, a["band_w"] + (t * N + n0) * 16)
        else:
            put64(gen.P_OUT, a["out"] + (t * N + n0) * 8)
            P[gen.P_CTN] = T * N * 8
            put64(gen.P_W, int(np.float64(a["wscalar"][t]).view(np.uint64)))
        P[gen.P_WLDS] = PARAM_BYTES + NCONS * 16 * TPITCH + w * 1024
        put64(gen.P_CID, a["order"] + (g * CG + w * NCH) * 4)
        put64(gen.P_PART, a["partial"] + (t * a["ntile"] + tile) * 8)
        P[gen.P_PCS] = T * a["ntile"] * 8
        P[gen.P_NVALID] = min(64, N - n0)
        P[gen.P_TRB] = PARAM_BYTES + w * 16 * TPITCH
    else:
        ll = w - NCONS
        put64(gen.PL_LT, a["ltab"] + (((gt * (smax + 3)) * NLOAD + ll) * LTABDW) * 4)
        put64(gen.PL_GROW, a["G"] + ((t * a["rows_per_target"]) * N + n0) * 8)
        P[gen.PL_ROWB] = N * 8
        P[gen.PL_RB0] = PARAM_BYTES
        P[gen.PL_BUFB] = a["cap"] * 512
        P[gen.PL_NSTEP] = nsteps
        P[gen.PL_NLANES] = min(32, (N - n0 + 1) // 2)
        P[gen.PL_NVAR] = a.get("nvar", 1)
        put64(gen.PL_G1, a.get("G1", a["G"]) + ((t * a["rows_per_target"]) * N + n0) * 8)
        put64(gen.PL_G2, a.get("G2", a["G"]) + ((t * a["rows_per_target"]) * N + n0) * 8)
    return P


def lds_bytes(cap):
    return PARAM_BYTES + max(3 * cap * 512, NCONS * 16 * TPITCH + NCONS * 1024)
