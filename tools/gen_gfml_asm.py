#!/usr/bin/env python
"""
gen_gfml_asm.py -- writes beat_amd/csrc/gfml_asm.inc: the consumer wavefront program of
k_gfstack_ml (beat_amd/csrc/gfcell.hip), the multilinear Green's-function stacking kernel for gfx950
with STATIC accumulators (round 4; the loader program is the one of tools/gen_gfcell_asm.py).

Reference arithmetic: beat/ffi/base.py:607-709 (multilinear branch :663-704) -- per (chain, target,
sample): acc = fma(G[row_k], w_k, acc) for the four corner rows k of the chain's (duration,
start-time) cell, patches in ascending order.  Same operations in the same order as k_gfstack
(gfstack.hip): bitwise equal.

Why another program: k_gfstack_cell selects a chain's accumulator through the VGPR index register;
every M0 write in that mode stalls the wavefront ~27 cycles and two thirds of its time is the
per-record / per-chain control skeleton.  Here NOTHING is dynamic except LDS
addresses:
  workgroup = (518-chain group, target, 64-sample tile) = 14 consumer + 2 loader wavefronts
  consumer  = 37 chains in a fixed order, lane <-> sample, accumulator of chain j = v[ACC+2j:ACC+2j+1]
  LDS rows  : DENSE layout per step: slot(d, s') = d*(S+1) + s', s' = s + 1, and s' = 0 holds a copy of the
      LAST start-time node of duration line d (the python negative-index wrap of base.py:513-517: floor node
      -1 -> S-1) when a chain needs it.  A chain's four rows then always sit at A, A+512 (floor-duration
      line: floor / ceil start time) and B, B+512 (ceil-duration line) -- two LDS addresses per chain,
      immediates for the rest; no per-row address arithmetic, no special cases.
  per chain : 2 x v_add_u32_dpp (A, B from the record), 4 x ds_read_b64 (contiguous 512-byte reads),
      4 x v_fmac_f64_dpp acc_j, w, x_k row_newbcast:(4q+k) -- accumulator and weight lane are
      immediates of the instruction; reads run one chain ahead of the FMAs.
  records   : per (wavefront, step) ceil(37/4) records of 256 B = 16 entries {weight f64, dword, pad}:
      entry 4q+k = weight k of chain q of the record, dword of entry 2q / 2q+1 = LDS byte offsets A / B of
      chain q.  One global_load_dwordx4 per record (lane l reads entry l mod 16: 16-lane replicated, what
      row_newbcast needs), four records ahead in a ring of five register sets -- vector loads return in
      order, so the wait is a constant s_waitcnt vmcnt(3).
One s_barrier per patch for all sixteen wavefronts (ring of three LDS row buffers, loaders two patches ahead).

    python tools/gen_gfml_asm.py        # rewrites beat_amd/csrc/gfml_asm.inc
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_gfcell_asm as base  # noqa: E402

e, lab, br, vp, sp, readlane = base.e, base.lab, base.br, base.vp, base.sp, base.readlane

NCHAIN = base.NCHAIN        # chains per consumer wavefront (the group structure of k_gfstack_cell)
NCONS, NLOAD = base.NCONS, base.NLOAD
REC = 256                   # bytes per record: 16 entries of 16 bytes
NREC = (NCHAIN + 3) // 4    # records per (wavefront, step)
WSTRIDE = NREC * REC        # bytes per (wavefront, step) in the record table
NRING = 5                   # register sets of the record ring (NREC % NRING == 0: ring position is static)
AHEAD = NRING - 1           # records requested ahead
assert NREC % NRING == 0 and (NREC + AHEAD) * REC < 4096   # 13-bit immediate offsets of global_load

# ---------------------------------------------------------------- consumer registers
V_IN = 0          # "%0": LDS address of the wavefront's parameter block
V_RING = 1        # lane*8 + LDS address of the row ring (epilogue: lane*8)
V_T0 = base.V_T0  # 4
V_PAR = base.V_PAR  # 5
V_AD = base.V_AD  # [6:9]: (A, B) of even chains, (A, B) of odd chains
V_L16 = 10        # (lane % 16) * 16: a lane's entry of a record
XA, XB = 12, 20   # row registers of even / odd chains: 4 pairs each
RREC = 28         # [28:47] record ring: set r = v[28+4r : 28+4r+3] (weight pair, dword, pad)
ACC = 48
V_LAST = ACC + 2 * NCHAIN - 1

S_NSTEP, S_WP, S_RB0 = base.S_NSTEP, base.S_WP, base.S_RB0
T0 = base.T0

ABL = set()       # timing experiments: 'nofma', 'nox', 'norec' (results are wrong with any of them)


def rec_w(i):
    return RREC + 4 * (i % NRING)


def request_record(i):
    """record i of the current step (i >= NREC: of the next step, the table is contiguous) -> ring set i % NRING"""
    if 'norec' in ABL or ('rec4' in ABL and i % 4 != 0 and i >= AHEAD):
        return
    r = rec_w(i)
    if 'x4' in ABL:
        e("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, V_L16, sp(S_WP), i * REC))
    else:
        e("global_load_dwordx3 v[%d:%d], v%d, %s offset:%d" % (r, r + 2, V_L16, sp(S_WP), i * REC))


def addresses(j):
    """LDS addresses A, B of chain j (record j // 4) -> address set j % 2"""
    i, q = j // 4, j % 4
    for h in range(2):
        e("v_add_u32_dpp v%d, v%d, v%d row_newbcast:%d row_mask:0xf bank_mask:0xf"
          % (V_AD + 2 * (j % 2) + h, rec_w(i) + 2, V_RING, 2 * q + h))


def reads(j):
    """the four rows of chain j in FMA order: (ceil d, ceil s), (ceil d, floor s), (floor d, ceil s), (floor d, floor s)"""
    if 'nox' in ABL:
        return
    x = XA if j % 2 == 0 else XB
    a, b = V_AD + 2 * (j % 2), V_AD + 2 * (j % 2) + 1
    e("ds_read_b64 %s, v%d offset:512" % (vp(x + 0), b))
    e("ds_read_b64 %s, v%d" % (vp(x + 2), b))
    e("ds_read_b64 %s, v%d offset:512" % (vp(x + 4), a))
    e("ds_read_b64 %s, v%d" % (vp(x + 6), a))


def fmas(j):
    if 'nofma' in ABL:
        return
    i, q = j // 4, j % 4
    x = XA if j % 2 == 0 else XB
    for k in range(4):
        e("v_fmac_f64_dpp %s, %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf"
          % (vp(ACC + 2 * j), vp(rec_w(i)), vp(x + 2 * k), 4 * q + k))


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
    e("s_barrier")                                     # rows of steps 0..2 in LDS
    e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))             # record 0
    addresses(0)
    reads(0)
    base._in_loop[0] = True
    lab("STEP")
    for j in range(NCHAIN):
        i, q = j // 4, j % 4
        if q == 0:
            # ring set (i + AHEAD) % NRING = set of record i - 1: its last reader was chain 4i - 1
            request_record(i + AHEAD)
        if j < NCHAIN - 1:
            jn = j + 1
            if jn % 4 == 0:
                e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))   # record jn // 4: the younger requests may be in flight
            addresses(jn)
            reads(jn)
            if 'nox' not in ABL:
                e("s_waitcnt lgkmcnt(4)")               # the rows of chain j (LDS returns in order)
            fmas(j)
        else:
            # last chain of the step
            e("s_sub_u32 s%d, s%d, 1" % (S_NSTEP, S_NSTEP))
            e("s_cmp_eq_u32 s%d, 0" % S_NSTEP)
            br("s_cbranch_scc1", "LASTSTEP")
            e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))       # record 0 of the next step
            e("s_add_u32 s%d, s%d, %d" % (S_WP, S_WP, WSTRIDE))
            e("s_addc_u32 s%d, s%d, 0" % (S_WP + 1, S_WP + 1))
            e("s_waitcnt lgkmcnt(0)")
            fmas(j)
            addresses(0)                                 # ring set 0 again: NREC % NRING == 0
            e("s_barrier")                               # rows of the next step published by the loaders
            reads(0)
            br("s_branch", "STEP")
    base._in_loop[0] = False
    lab("LASTSTEP")
    e("s_waitcnt lgkmcnt(0)")
    fmas(NCHAIN - 1)
    base.epilogue(XA, XB, ACC, NCHAIN, False)
    return list(L)


def clobbers():
    c = ["v%d" % i for i in range(1, V_LAST + 1)]
    c += ["s%d" % i for i in range(2, base.S_LAST + 1)]
    c += ["vcc", "m0", "scc", "memory"]
    return c


VARIANTS = [set(), {"nofma"}, {"nox"}, {"nofma", "nox"}, {"norec"}, {"nofma", "nox", "nodma"}, {"nofma", "nox", "rec4"},
            {"rec4"}, {"nodma"}, {"x4"}]


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "beat_amd", "csrc", "gfml_asm.inc")
    with open(out, "w") as f:
        f.write("// generated by tools/gen_gfml_asm.py -- do not edit\n")
        f.write("// the consumer wavefront program of k_gfstack_ml (see gfcell.hip and the generator)\n")
        for name, val in (("NCHAIN", NCHAIN), ("REC", REC), ("NREC", NREC), ("WSTRIDE", WSTRIDE), ("NVGPR", V_LAST + 1)):
            f.write("#define GM_%s %d\n" % (name, val))
        variants = VARIANTS if os.environ.get("GM_ABLATIONS") else VARIANTS[:1]
        f.write("#define GM_NVARIANT %d\n" % len(variants))
        cl = ", ".join('"%s"' % c for c in clobbers())
        for vi, abl in enumerate(variants):
            ABL.clear()
            ABL.update(abl)
            f.write("#define GM_CONSUMER_%d(PARAM_VGPR) asm volatile( \\\n" % vi)
            for line in consumer():
                f.write('    "%s\\n\\t" \\\n' % line)
            f.write('    : : "v"(PARAM_VGPR) : %s)\n' % cl)
        ABL.clear()
        if len(variants) > 1:
            base.ABL.add("nodma")
            cl = ", ".join('"%s"' % c for c in base.clobbers(base.LV_PAR))
            f.write("#define GM_LOADER_NODMA(PARAM_VGPR) asm volatile( \\\n")
            for line in base.loader(1):
                f.write('    "%s\\n\\t" \\\n' % line)
            f.write('    : : "v"(PARAM_VGPR) : %s)\n' % cl)
            base.ABL.clear()
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
