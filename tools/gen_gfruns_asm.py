#!/usr/bin/env python
"""
gen_gfruns_asm.py -- writes beat_amd/csrc/gfruns_asm.inc: the consumer wavefront program of
k_gfstack_runs (beat_amd/csrc/gfcell.hip), multilinear Green's-function stacking for gfx950 that reads the
four rows of a (duration, start-time) cell from LDS ONCE PER RUN of chains sharing the cell (round 4).

Reference arithmetic: beat/ffi/base.py:607-709 (multilinear branch :663-704) -- per (chain, target, sample):
acc = fma(G[row_k], w_k, acc) for the four corner rows k of the chain's cell, patches ascending.  Same operations
in the same order as k_gfstack (gfstack.hip): bitwise equal.

Where it comes from.  k_gfstack_ml (tools/gen_gfml_asm.py) walks a wavefront's 37 chains in a FIXED order and reads
four rows per chain: it is LDS-bound (row reads 77 % + LDS-DMA writes ~10 % of the LDS cycles at the 2.0 GHz the part
sustains; profiles/r4_variants.md).  k_gfstack_cell (round 3) shares row reads between chains but pays 33
instructions of control per batch record.  tools/micro/m0cost.hip showed that the index register itself is cheap
when the index comes from the SCALAR side (~2 cycles next to four FMAs, nothing at four waves per SIMD).  So:

  * per patch the chains of a wavefront are visited in CELL ORDER (k_gr_tables sorts them); the record stream keeps
    the static shape of k_gfstack_ml -- one 16-lane-replicated vector load per four chains: sixteen weights, the two
    LDS addresses (A, B) of each chain, and ONE packed dword per record: four 7-bit accumulator offsets + four
    "this chain opens a new cell" bits, fetched by one v_readlane per four chains and unpacked by the scalar unit;
  * the accumulator of a chain is v[ACC + M0] (s_set_gpr_idx mode, DST_REL; M0 by s_bfe_u32 + s_set_gpr_idx_idx);
  * rows are read only when the NEXT chain opens a new cell: two copies of the chain loop ("streams") differ in
    which row register set holds the current cell; a chain that opens a cell sends the wavefront through an
    out-of-line block (two address adds, four ds_read_b64 into the other set, one chain ahead of their use) and
    into the other stream -- one untaken scalar branch per chain that stays in its cell, two taken per new cell.
LDS row layout, loader wavefronts, row ring, barrier per patch and epilogues are those of k_gfstack_ml.

    python tools/gen_gfruns_asm.py        # rewrites beat_amd/csrc/gfruns_asm.inc
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_gfcell_asm as base  # noqa: E402
import gen_gfml_asm as ml  # noqa: E402

e, lab, br, vp, sp, readlane = base.e, base.lab, base.br, base.vp, base.sp, base.readlane

NCHAIN, NREC, REC, WSTRIDE, NRING, AHEAD = ml.NCHAIN, ml.NREC, ml.REC, ml.WSTRIDE, ml.NRING, ml.AHEAD
V_RING, V_T0, V_PAR, V_AD, V_L16, XA, XB, RREC, ACC, V_LAST = (ml.V_RING, ml.V_T0, ml.V_PAR, ml.V_AD, ml.V_L16, ml.XA,
                                                               ml.XB, ml.RREC, ml.ACC, ml.V_LAST)
S_NSTEP, S_WP, S_RB0 = ml.S_NSTEP, ml.S_WP, ml.S_RB0
S_ZERO = 3            # 0: accumulator offset of every instruction that is not a chain's FMA
S_PK = 8              # [8:9] packed dword of the record in use / of the next record (record i -> s[8 + i % 2])
S_IDX = 16            # scratch: the accumulator offset on its way to M0
PK_LANE = 8           # entry of a record whose dword holds the packed accumulator offsets / new-cell bits
PK_NEW = 28           # bit 28 + q: chain q of the record opens a new cell; bits [7q, 7q + 7): 2 x accumulator slot

ABL = set()           # timing experiments: 'nofma', 'nox'


def rec_w(i):
    return RREC + 4 * (i % NRING)


def idx0():
    e("s_set_gpr_idx_idx s%d" % S_ZERO)


def request_record(i):
    r = rec_w(i)
    e("global_load_dwordx3 v[%d:%d], v%d, %s offset:%d" % (r, r + 2, V_L16, sp(S_WP), i * REC))


def fetch_packed(i):
    """packed dword of record i -> its scalar register (index 0 must be in force)"""
    e("v_readlane_b32 s%d, v%d, %d" % (S_PK + i % 2, rec_w(i) + 2, PK_LANE))


def addresses(j, xset):
    i, q = j // 4, j % 4
    for h in range(2):
        e("v_add_u32_dpp v%d, v%d, v%d row_newbcast:%d row_mask:0xf bank_mask:0xf"
          % (V_AD + 2 * xset + h, rec_w(i) + 2, V_RING, 2 * q + h))


def reads(xset):
    if 'nox' in ABL:
        return
    x = XA if xset == 0 else XB
    a, b = V_AD + 2 * xset, V_AD + 2 * xset + 1
    e("ds_read_b64 %s, v%d offset:512" % (vp(x + 0), b))
    e("ds_read_b64 %s, v%d" % (vp(x + 2), b))
    e("ds_read_b64 %s, v%d offset:512" % (vp(x + 4), a))
    e("ds_read_b64 %s, v%d" % (vp(x + 6), a))


def fmas(j, xset):
    """chain at sorted position j: accumulator through M0, rows from set xset"""
    i, q = j // 4, j % 4
    e("s_bfe_u32 s%d, s%d, 0x%x" % (S_IDX, S_PK + i % 2, (7 << 16) | (7 * q)))
    e("s_set_gpr_idx_idx s%d" % S_IDX)
    if 'nofma' in ABL:
        return
    x = XA if xset == 0 else XB
    for k in range(4):
        e("v_fmac_f64_dpp %s, %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf"
          % (vp(ACC), vp(rec_w(i)), vp(x + 2 * k), 4 * q + k))


def block(r, p, out_of_line):
    """chain at sorted position r with its cell's rows in set p"""
    i, q = r // 4, r % 4
    lab("B%d_%d" % (r, p))
    if q == 0:
        idx0()                                  # (vector memory instructions are issued with offset 0 as well)
        request_record(i + AHEAD)
    if r < NCHAIN - 1:
        rn = r + 1
        if rn % 4 == 0:
            e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))
            if q != 0:
                idx0()
            fetch_packed(rn // 4)
        e("s_bitcmp1_b32 s%d, %d" % (S_PK + (rn // 4) % 2, PK_NEW + rn % 4))
        br("s_cbranch_scc1", "N%d_%d" % (r, p))
        if 'nox' not in ABL:
            e("s_waitcnt lgkmcnt(0)")               # (reads of this cell may still be in flight right after it opened)
        fmas(r, p)
        # falls through into B{r+1}_{p}

        def new_arm():
            lab("N%d_%d" % (r, p))
            idx0()
            addresses(rn, 1 - p)
            reads(1 - p)
            if 'nox' not in ABL:
                e("s_waitcnt lgkmcnt(4)")
            fmas(r, p)
            br("s_branch", "B%d_%d" % (rn, 1 - p))
        out_of_line.append(new_arm)
    else:
        # last chain of the step; chain 0 of the next step always opens a cell (other rows)
        e("s_sub_u32 s%d, s%d, 1" % (S_NSTEP, S_NSTEP))
        e("s_cmp_eq_u32 s%d, 0" % S_NSTEP)
        br("s_cbranch_scc1", "LAST_%d" % p)
        e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))       # record 0 of the next step
        e("s_add_u32 s%d, s%d, %d" % (S_WP, S_WP, WSTRIDE))
        e("s_addc_u32 s%d, s%d, 0" % (S_WP + 1, S_WP + 1))
        if q != 0:
            idx0()
        fetch_packed(NREC)                           # = record 0 of the next step (NREC even: same scalar register)
        addresses(0, 1 - p)
        if 'nox' not in ABL:
            e("s_waitcnt lgkmcnt(0)")
        fmas(r, p)
        idx0()
        e("s_barrier")                               # rows of the next step published by the loaders
        reads(1 - p)
        br("s_branch", "B0_%d" % (1 - p))


def consumer():
    L = base.L
    del L[:]
    base.lane_setup()
    e("v_lshlrev_b32 v%d, 3, v%d" % (V_RING, V_T0))
    e("v_and_b32 v%d, 15, v%d" % (V_L16, V_T0))
    e("v_lshlrev_b32 v%d, 4, v%d" % (V_L16, V_L16))
    base.read_params()
    for sreg, k in ((S_WP, base.P_WP), (S_WP + 1, base.P_WP + 1), (S_RB0, base.P_RB0), (S_NSTEP, base.P_NSTEP)):
        readlane(sreg, k)
    e("s_nop 4")
    e("v_add_u32 v%d, s%d, v%d" % (V_RING, S_RB0, V_RING))
    for r in range(AHEAD):
        request_record(r)
    for j in range(NCHAIN):
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j))
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j + 1))
    e("s_mov_b32 s%d, 0" % S_ZERO)
    e("s_barrier")                                     # rows of steps 0..2 in LDS
    # VGPR index mode (DST_REL) for the whole loop: v_fmac_f64_dpp v[ACC + M0[7:0]]; everything else runs with M0[7:0] = 0
    e("s_set_gpr_idx_on s%d, 0x8" % S_ZERO)
    e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))             # record 0
    fetch_packed(0)
    addresses(0, 0)
    reads(0)
    base._in_loop[0] = True
    assert NREC % 2 == 0
    ool = []
    for p in (0, 1):
        for r in range(NCHAIN):
            block(r, p, ool)
    for fn in ool:
        fn()
    base._in_loop[0] = False
    for p in (0, 1):
        lab("LAST_%d" % p)
        if 'nox' not in ABL:
            e("s_waitcnt lgkmcnt(0)")
        fmas(NCHAIN - 1, p)
        br("s_branch", "EPI")
    base.epilogue(XA, XB, ACC, NCHAIN, True)
    return list(L)


VARIANTS = [set(), {"nofma"}, {"nox"}, {"nofma", "nox"}]


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "beat_amd", "csrc", "gfruns_asm.inc")
    with open(out, "w") as f:
        f.write("// generated by tools/gen_gfruns_asm.py -- do not edit\n")
        f.write("// the consumer wavefront program of k_gfstack_runs (see gfcell.hip and the generator)\n")
        for name, val in (("PK_LANE", PK_LANE), ("PK_NEW", PK_NEW)):
            f.write("#define GR_%s %d\n" % (name, val))
        variants = VARIANTS if os.environ.get("GR_ABLATIONS") else VARIANTS[:1]
        f.write("#define GR_NVARIANT %d\n" % len(variants))
        cl = ", ".join('"%s"' % c for c in ml.clobbers())
        for vi, abl in enumerate(variants):
            ABL.clear()
            ABL.update(abl)
            f.write("#define GR_CONSUMER_%d(PARAM_VGPR) asm volatile( \\\n" % vi)
            for line in consumer():
                f.write('    "%s\\n\\t" \\\n' % line)
            f.write('    : : "v"(PARAM_VGPR) : %s)\n' % cl)
        ABL.clear()
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
