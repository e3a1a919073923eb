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

  * per patch the chains of a wavefront are visited in CELL ORDER (k_gm_tables<1> sorts them);
  * the record stream carries ONLY the weights: one 16-lane-replicated global_load_dwordx4 per EIGHT chains (entry
    l mod 16 of a 256-byte record pair = {weight l mod 16 of the first four chains, of the second four}: what
    row_newbcast needs), in a ring of five register quads, four pairs = 32 chains ahead.  The records come from the
    far side of the L2 (7 MB per call, read by 4096 workgroups) while the loaders pull 3 TB/s through it: with 16
    chains ahead the kernel took 13.1 ms, with 8 ahead 14.6 ms (profiles/r4_variants.md).  What the records cost is
    their instructions (a load and a wait), not their bytes (a build fetching with 16 lanes is not faster): hence
    eight chains per load;
  * TWO DWORDS PER CHAIN in scalar registers (scalar loads from a second table, one 80-dword line per wavefront and
    step): d = 0x4000 | accumulator slot | "the NEXT chain opens a new cell" << 31 and the LDS slots of the chain's two
    row pairs (A | B << 16).  A single scalar instruction, s_add_u32 m0, d, d, both selects the accumulator
    (M0 = 0x8000 | 2 x slot: s_set_gpr_idx mode, DST_REL, v_fmac_f64_dpp v[ACC + M0[7:0]]) and puts the new-cell bit
    into SCC.  tools/micro/chaincost.hip: with four wavefronts per SIMD a chain that stays in its cell costs 8.3 ns
    per SIMD as four bare FMAs, 10.2 ns with the three scalar instructions of the first version (s_bitcmp1 on a
    packed dword fetched by v_readlane, s_bfe_u32, s_set_gpr_idx_idx) and 9.3-9.7 ns with this one;
  * rows are read only when the NEXT chain opens a new cell: behind the chain's own FMAs, into the SAME eight row
    registers (two v_mad_u32_u16 on the halves of the slot dword, four ds_read_b64, s_waitcnt) -- one short scalar branch per chain
    that stays in its cell, nothing out of line.  (The first versions kept two row sets and two copies of the loop, so
    that the reads could be issued before the FMAs and waited for behind them: ablation 'two'.  It is 0-2.5 % slower:
    what counts is instructions and branches per wavefront, not the cover of an LDS round trip;
    tools/micro/chaincost.hip: a taken short branch costs what an untaken one costs.)  No s_waitcnt on the path of a
    chain that stays in its cell;
  * the descriptor registers are reloaded in two halves for the next step as soon as the chains that own them are
    done (chains 0-18 at chain 19; chains 19-36 right behind the barrier of the next step); s_waitcnt lgkmcnt(0)
    in every new-cell block, at the end of a step and at chain 12 keeps every consumer behind its load on every
    path (scalar loads return out of order: only a full wait counts).
LDS row layout, loader wavefronts, row ring, barrier per patch and epilogues are those of k_gfstack_ml.

    python tools/gen_gfruns_asm.py        # rewrites beat_amd/csrc/gfruns_asm.inc
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_gfcell_asm as base  # noqa: E402
import gen_gfml_asm as ml  # noqa: E402

e, lab, br, vp, sp, readlane = base.e, base.lab, base.br, base.vp, base.sp, base.readlane

NCHAIN, NREC = ml.NCHAIN, ml.NREC
NPAIR = NREC // 2     # a vector load brings the weights of EIGHT chains (two records of four): half the loads and waits
PAIR = 256            # bytes per record pair: sixteen entries {weight of record 2p, weight of record 2p + 1}
WSTRIDE = NPAIR * PAIR  # bytes per (wavefront, step) in the record table
NRING = NPAIR         # register quads of the record ring (ring position = pair index: static)
AHEAD = NRING - 1     # record pairs requested ahead (32 chains)
assert NREC % 2 == 0 and (NPAIR + AHEAD) * PAIR < 4096      # 13-bit immediate offsets of global_load
V_RING, V_T0, V_PAR, V_AD, V_L16, XA, XB, RREC, ACC, V_LAST = (ml.V_RING, ml.V_T0, ml.V_PAR, ml.V_AD, ml.V_L16, ml.XA,
                                                               ml.XB, ml.RREC, ml.ACC, ml.V_LAST)
assert RREC + 4 * NRING <= ACC
S_NSTEP, S_WP, S_RB0 = ml.S_NSTEP, ml.S_WP, ml.S_RB0
T0, T1 = base.T0, base.T1
S_ZERO = 3            # 0: accumulator offset of every instruction that is not a chain's FMA
S_DP = 6              # [6:7] the wavefront's descriptor line of the step in hand
# chain descriptors, two dwords per chain: chains 0..NHALF-1 in s[D_A ..], the rest in s[D_B ..]
NHALF = 19
D_A, D_B = 20, 60
DLINE = 80            # dwords per (wavefront, step) of the descriptor table: chain r at dwords 2r, 2r+1 (r < NHALF) or
DHALF = 40            # DHALF + 2(r - NHALF), +1
DSTRIDE = DLINE * 4
D_BASE = 0x4000       # d = D_BASE | slot | new << 31;  d + d = 0x8000 (DST_REL) | 2 * slot, carry = new
FORCE_WAIT = 12       # chain whose block waits for everything in flight (the second half of the descriptors)
assert D_A + 2 * NHALF <= D_B and D_B + 2 * (NCHAIN - NHALF) - 1 <= base.S_LAST

ABL = set()           # timing experiments: 'nofma', 'nox', 'nonew', 'norec', 'ahead2', 'ahead4' (wrong results except ahead*)


def rec_w(i):
    """register pair with the sixteen weights of record i (four chains)"""
    return RREC + 4 * ((i // 2) % NRING) + 2 * (i % 2)


def dreg(r):
    return D_A + 2 * r if r < NHALF else D_B + 2 * (r - NHALF)


def ddword(r):
    return 2 * r if r < NHALF else DHALF + 2 * (r - NHALF)


def idx0():
    e("s_set_gpr_idx_idx s%d" % S_ZERO)


def request_pair(p):
    r = RREC + 4 * (p % NRING)
    if 'norec' in ABL and base._in_loop[0]:
        return
    if 'rec16' in ABL and base._in_loop[0]:
        # timing only: 16 lanes fetch (a quarter of the bytes through the texture path; rows 1-3 of the weights stale)
        e("s_mov_b64 exec, 0xffff")
        e("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, V_L16, sp(S_WP), p * PAIR))
        e("s_mov_b64 exec, -1")
        return
    e("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, V_L16, sp(S_WP), p * PAIR))


def load_descriptors(first, count, byte_off):
    """scalar loads of the descriptor dwords of `count` chains from chain `first` on"""
    reg, dw, n_left = dreg(first), ddword(first), 2 * count
    while n_left:
        n = 16
        while n > n_left or reg % min(n, 4):
            n //= 2
        dst = "s%d" % reg if n == 1 else "s[%d:%d]" % (reg, reg + n - 1)
        e("s_load_dword%s %s, %s, 0x%x" % ("" if n == 1 else "x%d" % n, dst, sp(S_DP), byte_off + 4 * dw))
        reg, dw, n_left = reg + n, dw + n, n_left - n


V_C512 = 2            # 512: bytes per LDS row slot (v_mad_u32_u16 takes one scalar operand only)


def addresses(j, xset):
    """LDS addresses of the two row pairs of the chain at sorted position j (index 0 must be in force): slot halves of
    the chain's second descriptor dword x 512 + the lane's address in the row ring -- v_mad_u32_u16 picks the half
    itself (no scalar unpacking)"""
    e("v_mad_u32_u16 v%d, s%d, v%d, v%d" % (V_AD + 2 * xset, dreg(j) + 1, V_C512, V_RING))
    e("v_mad_u32_u16 v%d, s%d, v%d, v%d op_sel:[1,0,0,0]" % (V_AD + 2 * xset + 1, dreg(j) + 1, V_C512, V_RING))


def reads(xset):
    if 'nox' in ABL:
        return
    x = XA if xset == 0 else XB
    a, b = V_AD + 2 * xset, V_AD + 2 * xset + 1
    # (one ds_read2_b64 per row pair -- offset0:64 offset1:0 -- halves the LDS instructions of a new cell and is 2 %
    # SLOWER: 13.6 against 13.3 ms in one session)
    e("ds_read_b64 %s, v%d offset:512" % (vp(x + 0), b))
    e("ds_read_b64 %s, v%d" % (vp(x + 2), b))
    e("ds_read_b64 %s, v%d offset:512" % (vp(x + 4), a))
    e("ds_read_b64 %s, v%d" % (vp(x + 6), a))


def lgkm0():
    if 'nox' not in ABL:
        e("s_waitcnt lgkmcnt(0)")


def select(r):
    """accumulator of the chain at sorted position r -> M0; SCC = the next chain opens a cell"""
    e("s_add_u32 m0, s%d, s%d" % (dreg(r), dreg(r)))


def fma4(r, xset):
    if 'nofma' in ABL:
        return
    i, q = r // 4, r % 4
    x = XA if xset == 0 else XB
    for k in range(4):
        e("v_fmac_f64_dpp %s, %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf"
          % (vp(ACC), vp(rec_w(i)), vp(x + 2 * k), 4 * q + k))


def block(r, p, out_of_line):
    """chain at sorted position r with its cell's rows (landed) in set p"""
    i, q = r // 4, r % 4
    lab("B%d_%d" % (r, p))
    if r % 8 == 0:
        request_pair(r // 8 + AHEAD)
    if r == FORCE_WAIT:
        e("s_waitcnt lgkmcnt(0)")               # descriptors of chains NHALF.. (requested at the step's start)
    if r == NHALF:
        load_descriptors(0, NHALF, DSTRIDE)     # chains 0..NHALF-1 of the NEXT step: their registers are free
    if r < NCHAIN - 1:
        rn = r + 1
        if rn % 8 == 0:
            e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))     # record pair of chain rn
        select(r)
        if 'nonew' not in ABL:
            br("s_cbranch_scc1", "N%d_%d" % (r, p))
        fma4(r, p)
        # falls through into B{r+1}_{p}

        def new_arm():
            lab("N%d_%d" % (r, p))
            idx0()
            addresses(rn, 1 - p)
            reads(1 - p)
            select(r)
            fma4(r, p)
            lgkm0()                              # the new rows (and whatever scalar load is in flight)
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
        e("s_add_u32 s%d, s%d, %d" % (S_DP, S_DP, DSTRIDE))
        e("s_addc_u32 s%d, s%d, 0" % (S_DP + 1, S_DP + 1))
        select(r)
        fma4(r, p)
        e("s_waitcnt lgkmcnt(0)")                    # descriptors of chains 0..NHALF-1 of the next step
        idx0()
        addresses(0, 1 - p)
        e("s_barrier")                               # rows of the next step published by the loaders
        reads(1 - p)
        lgkm0()
        load_descriptors(NHALF, NCHAIN - NHALF, 0)   # chains NHALF.. of the step that starts (S_DP has moved on)
        br("s_branch", "B0_%d" % (1 - p))


def block_one(r):
    """chain at sorted position r; ONE copy of the chain loop and one row set: a chain whose successor opens a cell
    reads the new rows into the same registers right behind its own FMAs and waits for them (no cover for the LDS round
    trip, but no out-of-line block, no taken far branches, half the code: 1-2.5 % faster than the two-copy version,
    ablation 'two')"""
    i, q = r // 4, r % 4
    lab("B%d_0" % r)
    if r % 8 == 0:
        # (the VGPR index applies to vector ALU destinations only: a build with s_set_gpr_idx_idx 0 in front of this
        # load gives the same bits and is 0.8 % slower)
        request_pair(r // 8 + AHEAD)
    if r == FORCE_WAIT:
        e("s_waitcnt lgkmcnt(0)")               # descriptors of chains NHALF.. (requested at the step's start)
    if r == NHALF:
        load_descriptors(0, NHALF, DSTRIDE)     # chains 0..NHALF-1 of the NEXT step: their registers are free
    if r < NCHAIN - 1:
        rn = r + 1
        if rn % 8 == 0:
            e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))     # record pair of chain rn
        select(r)
        fma4(r, 0)
        br("s_cbranch_scc0", "B%d_0" % rn)          # the next chain stays in the cell
        idx0()
        addresses(rn, 0)
        reads(0)
        lgkm0()
    else:
        e("s_sub_u32 s%d, s%d, 1" % (S_NSTEP, S_NSTEP))
        e("s_cmp_eq_u32 s%d, 0" % S_NSTEP)
        br("s_cbranch_scc1", "LAST_0")
        e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))
        e("s_add_u32 s%d, s%d, %d" % (S_WP, S_WP, WSTRIDE))
        e("s_addc_u32 s%d, s%d, 0" % (S_WP + 1, S_WP + 1))
        e("s_add_u32 s%d, s%d, %d" % (S_DP, S_DP, DSTRIDE))
        e("s_addc_u32 s%d, s%d, 0" % (S_DP + 1, S_DP + 1))
        select(r)
        fma4(r, 0)
        e("s_waitcnt lgkmcnt(0)")
        idx0()
        addresses(0, 0)
        e("s_barrier")
        reads(0)
        lgkm0()
        load_descriptors(NHALF, NCHAIN - NHALF, 0)
        br("s_branch", "B0_0")


def consumer():
    L = base.L
    del L[:]
    base.lane_setup()
    e("v_lshlrev_b32 v%d, 3, v%d" % (V_RING, V_T0))
    e("v_and_b32 v%d, 15, v%d" % (V_L16, V_T0))
    e("v_lshlrev_b32 v%d, 4, v%d" % (V_L16, V_L16))    # (lane % 16) * 16: a lane's entry of a record pair
    base.read_params()
    for sreg, k in ((S_WP, base.P_WP), (S_WP + 1, base.P_WP + 1), (S_RB0, base.P_RB0), (S_NSTEP, base.P_NSTEP),
                    (S_DP, base.P_DP), (S_DP + 1, base.P_DP + 1)):
        readlane(sreg, k)
    e("s_nop 4")
    e("v_add_u32 v%d, s%d, v%d" % (V_RING, S_RB0, V_RING))
    e("v_mov_b32 v%d, 0x200" % V_C512)
    load_descriptors(0, NHALF, 0)
    load_descriptors(NHALF, NCHAIN - NHALF, 0)
    for r in range(AHEAD):
        request_pair(r)
    for j in range(NCHAIN):
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j))
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j + 1))
    e("s_mov_b32 s%d, 0" % S_ZERO)
    e("s_barrier")                                     # rows of steps 0..2 in LDS
    # VGPR index mode (DST_REL) for the whole loop: v_fmac_f64_dpp v[ACC + M0[7:0]]; everything else runs with M0[7:0] = 0
    e("s_set_gpr_idx_on s%d, 0x8" % S_ZERO)
    e("s_waitcnt vmcnt(%d) lgkmcnt(0)" % (AHEAD - 1))  # record 0, the descriptors of step 0
    addresses(0, 0)
    reads(0)
    e("s_waitcnt lgkmcnt(0)")                          # rows of chain 0
    base._in_loop[0] = True
    ool = []
    for p in ((0,) if 'two' not in ABL else (0, 1)):
        for r in range(NCHAIN):
            if 'two' not in ABL:
                block_one(r)
            else:
                block(r, p, ool)
    for fn in ool:
        fn()
    base._in_loop[0] = False
    for p in ((0,) if 'two' not in ABL else (0, 1)):
        lab("LAST_%d" % p)
        select(NCHAIN - 1)
        fma4(NCHAIN - 1, p)
        e("s_waitcnt lgkmcnt(0)")                      # the descriptor load ahead must not land in the epilogue's registers
        br("s_branch", "EPI")
    base.epilogue(XA, XB, ACC, NCHAIN, True)
    return list(L)


VARIANTS = [set(), {"nofma"}, {"nox"}, {"nonew"}, {"norec"}, {"nonew", "norec"}, {"nonew", "norec", "nofma"}, {"ahead2"}, {"two"}, {"rec16"}]


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "beat_amd", "csrc", "gfruns_asm.inc")
    with open(out, "w") as f:
        f.write("// generated by tools/gen_gfruns_asm.py -- do not edit\n")
        f.write("// the consumer wavefront program of k_gfstack_runs (see gfcell.hip and the generator)\n")
        for name, val in (("PAIR", PAIR), ("WSTRIDE", WSTRIDE), ("NHALF", NHALF), ("DLINE", DLINE), ("DHALF", DHALF),
                          ("D_BASE", D_BASE)):
            f.write("#define GR_%s %d\n" % (name, val))
        variants = VARIANTS if os.environ.get("GR_ABLATIONS") else VARIANTS[:1]
        f.write("#define GR_NVARIANT %d\n" % len(variants))
        cl = ", ".join('"%s"' % c for c in ml.clobbers())
        for vi, abl in enumerate(variants):
            ABL.clear()
            ABL.update(abl)
            global AHEAD
            AHEAD = 2 if "ahead2" in abl else NRING - 1
            f.write("#define GR_CONSUMER_%d(PARAM_VGPR) asm volatile( \\\n" % vi)
            for line in consumer():
                f.write('    "%s\\n\\t" \\\n' % line)
            f.write('    : : "v"(PARAM_VGPR) : %s)\n' % cl)
        ABL.clear()
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
