#!/usr/bin/env python
"""
gen_gfruns_asm.py -- writes beat_amd/csrc/gfruns_asm.inc: the two wavefront programs of k_gfstack_runs
(beat_amd/csrc/gfcell.hip), the multilinear Green's-function stacking kernel for gfx950 that reads the four rows
of a (duration, start-time) cell from LDS ONCE PER RUN of chains sharing the cell (round 4) and, since round 5,
stages a patch whose chain group touches more rows than LDS holds in several ROW PASSES.

Reference arithmetic: beat/ffi/base.py:607-709 (multilinear branch :663-704) -- per (chain, target, sample):
acc = fma(G[row_k], w_k, acc) for the four corner rows k of the chain's cell, patches ascending.  Same operations
in the same order as k_gfstack (gfstack.hip): bitwise equal.

Mapping: workgroup = (518-chain group, target, 64-sample tile) = 14 consumer + 2 loader wavefronts; one s_barrier
per step for all sixteen.  A STEP is one (patch, row pass, slip variable); the table kernels (k_gm_*) write, per
step, the loaders' request lines and the consumers' descriptor lines and weight records.

  loader:   LDS-DMA (global_load_lds_dwordx4) of every row segment the step needs into a ring of three LDS row
      buffers, two steps ahead.  A request line (48 dwords per loader and step) = count, first library row of the
      step's patch, 46 requests: rowA (16 bits, relative to that row) | (rowB - rowA) << 16 (8 bits; 0 = a single
      row) | LDS slot of rowA << 24; the two rows of a pair land in adjacent slots (lanes 0-31 move rowA, lanes
      32-63 rowB).
  consumer: 37 chains, lane <-> sample, accumulator of chain j = v[ACC+2j : ACC+2j+1], selected through the VGPR
      index register (s_set_gpr_idx_on, DST_REL).  Per step the wavefront walks 37 POSITIONS; the table kernel
      puts the chains that take part in the step (all of them when a patch is one pass) at positions 0..n-1 in
      CELL ORDER, the rest are PADS: zero weights into the scratch accumulator (slot 37), no row reads.
      * the record stream carries ONLY the weights: one 16-lane-replicated global_load_dwordx4 per EIGHT
        positions (entry l mod 16 of a 256-byte record pair = {weight l mod 16 of the first four, of the second
        four}: what row_newbcast needs), in a ring of five register quads, four pairs = 32 positions ahead;
      * TWO DWORDS PER POSITION in scalar registers (scalar loads, one 80-dword line per wavefront and step):
        d = 0x4000 | accumulator slot | "the NEXT position opens a new cell" << 31 and the LDS slots of the chain's
        two row pairs (A | B << 16).  A single scalar instruction, s_add_u32 m0, d, d, both selects the accumulator
        (M0 = 0x8000 | 2 x slot) and puts the new-cell bit into SCC (tools/micro/chaincost.hip);
      * rows are read only when the NEXT position opens a new cell: behind the chain's own FMAs, into the SAME
        eight row registers (two v_mad_u32_u16 on the halves of the slot dword, four ds_read_b64, s_waitcnt);
      * the descriptor registers are reloaded in two halves for the next step as soon as the positions that own
        them are done; s_waitcnt lgkmcnt(0) in every new-cell block, at the end of a step and at position 12
        keeps every consumer behind its load on every path (scalar loads return out of order).
  LDS rows: a step's buffer holds the rows of its pass COMPACTLY in (duration line, start-time node) order -- the
      four rows of a chain are A, A+512 (floor-duration line: floor / ceil start time) and B, B+512 (ceil-duration
      line); the floor node of start-time node 0 is a copy of the line's LAST node (python negative-index wrap,
      base.py:513-517), staged only when a chain needs it.

The programs are generated because they are register-allocated by hand (the accumulators must be a contiguous
physical VGPR range) and unrolled; tools/gfcell_emu.py interprets them on the CPU (tests/test_gfcell_program.py).
Rounds 3 / 4 shipped two more consumer programs (k_gfstack_cell: batch records; k_gfstack_ml: static accumulators,
LDS-bound); both were retired in round 5 (docs/design_history.md 3.1d-e keep their measurements).

    python tools/gen_gfruns_asm.py        # rewrites beat_amd/csrc/gfruns_asm.inc
"""
import os

NCHAIN = 37            # chains per consumer wavefront
NCONS, NLOAD = 14, 2   # consumer / loader wavefronts
if os.environ.get("GR_NCONS"):       # experiments (tools/build_ablations.sh): other splits of the sixteen wavefronts
    NCONS = int(os.environ["GR_NCONS"])
    NLOAD = 16 - NCONS
NPOS = NCHAIN          # positions a consumer walks per step
SCRATCH = NCHAIN       # accumulator slot of the pads
LREQ = 46              # row requests per loader and step
LTABDW = 48            # dwords per (loader, step) in the request table: count, first row of the patch, requests
LTAB = LTABDW * 4

NREC = (NCHAIN + 3) // 4    # records of four positions per (wavefront, step)
NPAIR = NREC // 2     # a vector load brings the weights of EIGHT positions (two records of four)
PAIR = 256            # bytes per record pair: sixteen entries {weight of record 2p, weight of record 2p + 1}
WSTRIDE = NPAIR * PAIR  # bytes per (wavefront, step) in the record table
NRING = NPAIR         # register quads of the record ring (ring position = pair index: static)
AHEAD = NRING - 1     # record pairs requested ahead (32 positions)
assert NREC % 2 == 0 and (NPAIR + AHEAD) * PAIR < 4096      # 13-bit immediate offsets of global_load

# ---------------------------------------------------------------- consumer registers
V_IN = 0          # "%0": LDS address of the wavefront's parameter block (allocated by the compiler)
V_RING = 1        # lane*8 + LDS address of the row ring (epilogue: lane*8)
V_C512 = 2        # 512: bytes per LDS row slot (v_mad_u32_u16 takes one scalar operand only)
V_T0 = 4
V_PAR = 5         # parameter dwords (lane k = dword k)
V_AD = 6          # [6:9] LDS addresses of the row pairs
V_L16 = 10        # (lane % 16) * 16: a lane's entry of a record pair
XA, XB = 12, 20   # row registers (XB: epilogue scratch): 4 pairs each
RREC = 28         # [28:47] record ring: quad r = v[28+4r : 28+4r+3]
ACC = 48
V_LAST = ACC + 2 * (NCHAIN + 1) - 1      # (+ the scratch accumulator)
assert RREC + 4 * NRING <= ACC and V_LAST < 128

S_NSTEP = 2
S_ZERO = 3            # 0: accumulator offset of every instruction that is not a chain's FMA
S_WP = 4
S_DP = 6              # [6:7] the wavefront's descriptor line of the step in hand
S_RB0 = 15
T0, T1, T2, T3 = 16, 17, 18, 19   # (T2:T3 is an aligned pair: addresses)
S_LAST = 95      # user SGPRs stop here: VCC, FLAT_SCRATCH and XNACK_MASK take the top six
# position descriptors, two dwords each: positions 0..NHALF-1 in s[D_A ..], the rest in s[D_B ..]
NHALF = 19
D_A, D_B = 20, 60
DLINE = 80            # dwords per (wavefront, step) of the descriptor table: position r at dwords 2r, 2r+1 (r < NHALF) or
DHALF = 40            # DHALF + 2(r - NHALF), +1
DSTRIDE = DLINE * 4
D_BASE = 0x4000       # d = D_BASE | slot | new << 31;  d + d = 0x8000 (DST_REL) | 2 * slot, carry = new
FORCE_WAIT = 12       # position whose block waits for everything in flight (the second half of the descriptors)
# EARLY EXIT (round 5): with row passes a wavefront has only some of its chains in a step (12 of 37 on the tutorial grid);
# the table kernel puts them at positions 0..n-1 and n into dword D_NCH of the descriptor line.  At the checkpoints below
# the walk leaves the step when every chain of the pass is done (the pads up to the checkpoint still run: zero weights
# into the scratch accumulator); the exit stub issues the loads the skipped blocks would have issued.
CHECKPOINTS = (4, 8, 12, 16, 24)
D_NCH = 38            # dword of the descriptor line that holds n (the two dwords behind the first half are free)
S_NCH, S_NCH_NEXT = 8, 9   # n of the step in hand / of the next step (loaded with the next step's first half)
assert D_A + 2 * NHALF <= D_B and D_B + 2 * (NPOS - NHALF) - 1 <= S_LAST

# parameter block of a consumer wavefront (dwords)
P_WP, P_RB0, P_NSTEP = 0, 4, 5
P_DP = 8
P_OUT, P_CTN, P_MODE, P_DATA, P_W, P_CID, P_PART, P_PCS, P_NVALID, P_TRB = 16, 18, 19, 20, 22, 24, 26, 28, 29, 30
P_WLDS = 31      # mode 3: LDS address of the wavefront's 1 KB for the tile's band rows (w0, w1)
# mode 3 (bidiagonal misfit, round 6) re-uses three words: P_OUT = edges + (t * ntile + tile) * 16 (chain stride 2 * PCS),
# P_W = address of the band rows of the tile's first sample, P_CTN = 1 when the tile ends the trace
# parameter block of a loader wavefront
PL_LT, PL_GROW, PL_ROWB, PL_RB0, PL_BUFB, PL_NSTEP, PL_NLANES = 0, 2, 5, 6, 7, 8, 9
PL_NVAR, PL_G1, PL_G2 = 10, 12, 14   # slip variables (steps cycle through their libraries), bases 2 and 3

L = []
ABL = set()   # timing experiments (GR_ABLATIONS builds; wrong results): 'nofma', 'nox', 'nobar', 'nonew', 'norec', 'nodma'


def e(s):
    if 'nobar' in ABL and s == 's_barrier' and _in_loop[0]:
        return
    L.append(s)


_in_loop = [False]


def lab(name):
    e("%s_%%=:" % name)


def br(op, name):
    e("%s %s_%%=" % (op, name))


def vp(r):
    return "v[%d:%d]" % (r, r + 1)


def sp(r, n=2):
    return "s[%d:%d]" % (r, r + n - 1)


def readlane(sreg, k, v=V_PAR):
    e("v_readlane_b32 s%d, v%d, %s" % (sreg, v, k if isinstance(k, str) else "%d" % k))


def lane_setup():
    e("v_mbcnt_lo_u32_b32 v%d, -1, 0" % V_T0)
    e("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (V_T0, V_T0))


def read_params():
    e("v_lshlrev_b32 v%d, 2, v%d" % (V_PAR, V_T0))
    e("v_add_u32 v%d, v%d, %%0" % (V_PAR, V_PAR))
    e("ds_read_b32 v%d, v%d" % (V_PAR, V_PAR))
    e("s_waitcnt lgkmcnt(0)")



# =============================================================================== epilogue
def epilogue(XA, XB, ACC, NCHAIN, idx_mode):
    """the three epilogues of a consumer wavefront (synthetics | residual store | scalar-covariance misfit); shared
    by the cell program above and the static-accumulator program of tools/gen_gfml_asm.py.  XA, XB: eight free
    VGPRs each (the row registers), ACC: first accumulator (chain j = v[ACC+2j : ACC+2j+1])"""
    V_D = XA
    V_T1 = XA + 2
    V_T2 = XA + 4
    lab("EPI")
    if idx_mode:
        e("s_set_gpr_idx_off")
    e("s_waitcnt vmcnt(0)")                            # records requested beyond the last step
    e("s_barrier")                                     # every wavefront is done with the row ring
    S_OUT, S_CTN, S_MODE, S_DATA, S_W, S_CID, S_PART, S_PCS, S_NVAL, S_TRB = 22, 24, 25, 26, 28, 30, 84, 86, 87, 88
    S_WLDS = 89
    T4, T5 = 20, 21
    for sreg, k in ((S_OUT, P_OUT), (S_OUT + 1, P_OUT + 1), (S_CTN, P_CTN), (S_MODE, P_MODE),
                    (S_DATA, P_DATA), (S_DATA + 1, P_DATA + 1), (S_W, P_W), (S_W + 1, P_W + 1),
                    (S_CID, P_CID), (S_CID + 1, P_CID + 1), (S_PART, P_PART), (S_PART + 1, P_PART + 1),
                    (S_PCS, P_PCS), (S_NVAL, P_NVALID), (S_TRB, P_TRB), (S_WLDS, P_WLDS)):
        readlane(sreg, k)
    e("s_nop 4")
    CID = 32   # s[32:79]: chain ids of the accumulators (37 used)
    for k in range(3):
        e("s_load_dwordx16 %s, %s, 0x%x" % (sp(CID + 16 * k, 16), sp(S_CID), 64 * k))
    lane_setup()
    e("v_lshlrev_b32 v%d, 3, v%d" % (V_RING, V_T0))          # lane*8 from here on
    e("s_mov_b64 vcc, -1")                                   # samples of the tile inside the trace
    e("s_cmp_ge_u32 s%d, 64" % S_NVAL)
    br("s_cbranch_scc1", "FULL")
    e("s_bfm_b64 vcc, s%d, 0" % S_NVAL)
    lab("FULL")
    e("s_waitcnt lgkmcnt(0)")
    e("s_cmp_eq_u32 s%d, 0" % S_MODE)
    br("s_cbranch_scc1", "SYN")
    e("v_mov_b32 v%d, 0" % V_D)
    e("v_mov_b32 v%d, 0" % (V_D + 1))
    e("s_mov_b64 exec, vcc")
    e("global_load_dwordx2 %s, v%d, %s" % (vp(V_D), V_RING, sp(S_DATA)))
    e("s_waitcnt vmcnt(0)")
    e("s_mov_b64 exec, -1")
    e("s_cmp_eq_u32 s%d, 1" % S_MODE)
    br("s_cbranch_scc1", "SCAL")
    e("s_cmp_eq_u32 s%d, 3" % S_MODE)
    br("s_cbranch_scc1", "BAND")

    def store_loop(tag, resid):
        e("s_mov_b64 exec, vcc")
        for j in range(NCHAIN):
            e("s_cmp_eq_u32 s%d, -1" % (CID + j))
            br("s_cbranch_scc1", "SK%s%d" % (tag, j))
            e("s_mul_hi_u32 s%d, s%d, s%d" % (T1, CID + j, S_CTN))
            e("s_mul_i32 s%d, s%d, s%d" % (T0, CID + j, S_CTN))
            e("s_add_u32 s%d, s%d, s%d" % (T2, T0, S_OUT))
            e("s_addc_u32 s%d, s%d, s%d" % (T3, T1, S_OUT + 1))
            if resid:
                tmp = V_T1 if (j & 1) == 0 else V_T2
                e("v_add_f64 %s, %s, -%s" % (vp(tmp), vp(V_D), vp(ACC + 2 * j)))   # seismic.py:1332
                e("global_store_dwordx2 v%d, %s, %s" % (V_RING, vp(tmp), sp(T2)))
            else:
                e("global_store_dwordx2 v%d, %s, %s" % (V_RING, vp(ACC + 2 * j), sp(T2)))
            lab("SK%s%d" % (tag, j))
        e("s_mov_b64 exec, -1")
        br("s_branch", "END")

    store_loop("R", True)
    lab("SYN")
    store_loop("S", False)
    # ---- scalar-covariance misfit: partial[c, t, tile] = sum_i (w (d_i - syn_i))^2, i ascending:
    # 16 chains at a time through a transposed LDS tile (row pitch 65 doubles), lane <-> chain
    lab("SCAL")
    V_WA, V_RA, V_C, V_L4 = V_AD, V_AD + 1, V_AD + 2, V_AD + 3
    Q = XA + 6
    TPITCH = 65 * 8
    e("v_add_u32 v%d, s%d, v%d" % (V_WA, S_TRB, V_RING))
    e("v_lshrrev_b32 v%d, 3, v%d" % (V_T0, V_RING))
    e("v_mul_u32_u24 v%d, %d, v%d" % (V_RA, TPITCH, V_T0))
    e("v_add_u32 v%d, s%d, v%d" % (V_RA, S_TRB, V_RA))
    e("v_lshlrev_b32 v%d, 2, v%d" % (V_L4, V_T0))
    for r in range((NCHAIN + 15) // 16):
        n = min(16, NCHAIN - 16 * r)
        for jj in range(n):
            j = 16 * r + jj
            tmp = V_T1 if (jj & 1) == 0 else V_T2
            e("v_add_f64 %s, %s, -%s" % (vp(tmp), vp(V_D), vp(ACC + 2 * j)))
            e("v_mul_f64 %s, %s, %s" % (vp(tmp), sp(S_W), vp(tmp)))     # distributions.py:128 with W = w I
            e("v_cndmask_b32 v%d, 0, v%d, vcc" % (tmp, tmp))             # samples beyond N contribute 0
            e("v_cndmask_b32 v%d, 0, v%d, vcc" % (tmp + 1, tmp + 1))
            e("ds_write_b64 v%d, %s offset:%d" % (V_WA, vp(tmp), jj * TPITCH))
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v%d, 0" % Q)
        e("v_mov_b32 v%d, 0" % (Q + 1))
        e("s_mov_b64 exec, 0x%x" % ((1 << n) - 1))
        e("global_load_dword v%d, v%d, %s offset:%d" % (V_C, V_L4, sp(S_CID), 64 * r))
        for i0 in range(0, 64, 4):
            for k in range(4):
                e("ds_read_b64 %s, v%d offset:%d" % (vp(XB + 2 * k), V_RA, (i0 + k) * 8))
            e("s_waitcnt lgkmcnt(0)")
            for k in range(4):
                e("v_fma_f64 %s, %s, %s, %s" % (vp(Q), vp(XB + 2 * k), vp(XB + 2 * k), vp(Q)))
        e("s_waitcnt vmcnt(0)")
        e("v_mov_b32 v%d, s%d" % (V_T1, S_PART))
        e("v_mov_b32 v%d, s%d" % (V_T1 + 1, S_PART + 1))
        e("v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (vp(V_T2), sp(T2), V_C, S_PCS, vp(V_T1)))
        e("v_cmp_ne_u32 %s, -1, v%d" % (sp(T2), V_C))
        e("s_and_b64 exec, exec, %s" % sp(T2))
        e("global_store_dwordx2 %s, %s, off" % (vp(V_T2), vp(Q)))
        e("s_mov_b64 exec, -1")
        e("s_waitcnt vmcnt(0)")
    br("s_branch", "END")
    # ---- bidiagonal whitening operator (mode 3, round 6; distributions.py:119-138 with W[i,i], W[i,i+1] only): per
    # (chain, tile) in the CANONICAL order of quadform.hip -- q = sum_{i<63} y_i^2 ascending, y_i = fma(w1_i, r_{i+1},
    # fma(w0_i, r_i, 0)), + the trace's very last sample --, partial[c,t,tile] = q and edges[c,t,tile] = (first, last
    # residual): k_sum_tiles_band1 adds the term of a tile's last sample (its neighbour is the next tile's first residual).
    # The tile's band rows go to LDS once (lane i: (w0_i, w1_i), zeros beyond the trace), then 16 chains at a time through
    # the transposed tile as above: lane <-> chain, the weights read at a uniform address
    lab("BAND")
    V_WU, V_WL16 = 11, V_L16
    WQ = RREC                          # [WQ+4k : WQ+4k+3] = (w0, w1) of sample i0 + k
    Y = RREC + 16                      # y_i; [Y : Y+3] stages the lane's band row first
    RF, RI = V_T1, V_T2                # the tile's first residual, r_i (carried)
    A1, A2, A3 = XB, XB + 2, XB + 4    # address temporaries behind the sample loop
    e("v_lshlrev_b32 v%d, 4, v%d" % (V_WL16, V_T0))            # lane * 16 (V_T0: the lane, lane_setup above)
    for k in range(4):
        e("v_mov_b32 v%d, 0" % (Y + k))
    e("s_mov_b64 exec, vcc")
    e("global_load_dwordx4 v[%d:%d], v%d, %s" % (Y, Y + 3, V_WL16, sp(S_W)))
    e("s_waitcnt vmcnt(0)")
    e("s_mov_b64 exec, -1")
    e("v_add_u32 v%d, s%d, v%d" % (V_WU, S_WLDS, V_WL16))
    e("ds_write_b128 v%d, v[%d:%d]" % (V_WU, Y, Y + 3))
    e("v_mov_b32 v%d, s%d" % (V_WU, S_WLDS))
    e("s_lshl_b32 s%d, s%d, 1" % (T5, S_PCS))                  # chain stride of `edges`
    e("v_add_u32 v%d, s%d, v%d" % (V_WA, S_TRB, V_RING))
    e("v_lshrrev_b32 v%d, 3, v%d" % (V_T0, V_RING))
    e("v_mul_u32_u24 v%d, %d, v%d" % (V_RA, TPITCH, V_T0))
    e("v_add_u32 v%d, s%d, v%d" % (V_RA, S_TRB, V_RA))
    e("v_lshlrev_b32 v%d, 2, v%d" % (V_L4, V_T0))
    e("s_waitcnt lgkmcnt(0)")
    for r in range((NCHAIN + 15) // 16):
        n = min(16, NCHAIN - 16 * r)
        for jj in range(n):
            j = 16 * r + jj
            tmp = V_T1 if (jj & 1) == 0 else V_T2
            e("v_add_f64 %s, %s, -%s" % (vp(tmp), vp(V_D), vp(ACC + 2 * j)))      # seismic.py:1332
            e("v_cndmask_b32 v%d, 0, v%d, vcc" % (tmp, tmp))                         # samples beyond N: residual 0
            e("v_cndmask_b32 v%d, 0, v%d, vcc" % (tmp + 1, tmp + 1))
            e("ds_write_b64 v%d, %s offset:%d" % (V_WA, vp(tmp), jj * TPITCH))
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v%d, 0" % Q)
        e("v_mov_b32 v%d, 0" % (Q + 1))
        e("s_mov_b64 exec, 0x%x" % ((1 << n) - 1))
        e("global_load_dword v%d, v%d, %s offset:%d" % (V_C, V_L4, sp(S_CID), 64 * r))
        e("ds_read_b64 %s, v%d" % (vp(RI), V_RA))
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v%d, v%d" % (RF, RI))
        e("v_mov_b32 v%d, v%d" % (RF + 1, RI + 1))
        for i0 in range(0, 64, 4):
            for k in range(4):
                e("ds_read_b64 %s, v%d offset:%d" % (vp(WQ + 4 * k), V_WU, (i0 + k) * 16))
                e("ds_read_b64 %s, v%d offset:%d" % (vp(WQ + 4 * k + 2), V_WU, (i0 + k) * 16 + 8))
                if i0 + k < 63:
                    e("ds_read_b64 %s, v%d offset:%d" % (vp(XB + 2 * k), V_RA, (i0 + k + 1) * 8))
            e("s_waitcnt lgkmcnt(0)")
            last = RI
            for k in range(4):
                if i0 + k == 63:
                    break
                e("v_fma_f64 %s, %s, %s, 0" % (vp(Y), vp(WQ + 4 * k), vp(last)))
                e("v_fma_f64 %s, %s, %s, %s" % (vp(Y), vp(WQ + 4 * k + 2), vp(XB + 2 * k), vp(Y)))
                e("v_fma_f64 %s, %s, %s, %s" % (vp(Q), vp(Y), vp(Y), vp(Q)))
                last = XB + 2 * k
            e("v_mov_b32 v%d, v%d" % (RI, last))
            e("v_mov_b32 v%d, v%d" % (RI + 1, last + 1))
        # the trace's very last sample (no neighbour; beyond the trace the staged w0 is 0)
        e("s_cmp_eq_u32 s%d, 0" % S_CTN)
        br("s_cbranch_scc1", "BNE%d" % r)
        e("v_fma_f64 %s, %s, %s, 0" % (vp(Y), vp(WQ + 12), vp(RI)))
        e("v_fma_f64 %s, %s, %s, %s" % (vp(Q), vp(Y), vp(Y), vp(Q)))
        lab("BNE%d" % r)
        e("s_waitcnt vmcnt(0)")
        e("v_mov_b32 v%d, s%d" % (A1, S_PART))
        e("v_mov_b32 v%d, s%d" % (A1 + 1, S_PART + 1))
        e("v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (vp(A2), sp(T2), V_C, S_PCS, vp(A1)))
        e("v_mov_b32 v%d, s%d" % (A1, S_OUT))
        e("v_mov_b32 v%d, s%d" % (A1 + 1, S_OUT + 1))
        e("v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (vp(A3), sp(T2), V_C, T5, vp(A1)))
        e("v_cmp_ne_u32 %s, -1, v%d" % (sp(T2), V_C))
        e("s_and_b64 exec, exec, %s" % sp(T2))
        e("global_store_dwordx2 %s, %s, off" % (vp(A2), vp(Q)))
        e("global_store_dwordx2 %s, %s, off" % (vp(A3), vp(RF)))
        e("global_store_dwordx2 %s, %s, off offset:8" % (vp(A3), vp(RI)))
        e("s_mov_b64 exec, -1")
        e("s_waitcnt vmcnt(0)")
    lab("END")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")



# =============================================================================== consumer
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
    if 'norec' in ABL and _in_loop[0]:
        return
    e("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, V_L16, sp(S_WP), p * PAIR))


def load_descriptors(first, count, byte_off):
    """scalar loads of the descriptor dwords of `count` chains from chain `first` on"""
    reg, dw, n_left = dreg(first), ddword(first), 2 * count
    if 'nosm' in ABL and _in_loop[0]:    # (timing only: the descriptors of step 0 stay)
        return
    while n_left:
        n = 16
        while n > n_left or reg % min(n, 4):
            n //= 2
        dst = "s%d" % reg if n == 1 else "s[%d:%d]" % (reg, reg + n - 1)
        e("s_load_dword%s %s, %s, 0x%x" % ("" if n == 1 else "x%d" % n, dst, sp(S_DP), byte_off + 4 * dw))
        reg, dw, n_left = reg + n, dw + n, n_left - n


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
    if 'one' in ABL:     # timing only: one row and one FMA per chain (what a nearest-neighbour form of this program would do)
        return
    e("ds_read_b64 %s, v%d" % (vp(x + 2), b))
    e("ds_read_b64 %s, v%d offset:512" % (vp(x + 4), a))
    e("ds_read_b64 %s, v%d" % (vp(x + 6), a))


def lgkm0():
    if 'nox' not in ABL and 'nolgkm' not in ABL:      # ('nolgkm': the emulator's test that a missing wait is caught)
        e("s_waitcnt lgkmcnt(0)")


def select(r):
    """accumulator of the chain at sorted position r -> M0; SCC = the next chain opens a cell"""
    e("s_add_u32 m0, s%d, s%d" % (dreg(r), dreg(r)))


def fma4(r, xset):
    if 'nofma' in ABL:
        return
    i, q = r // 4, r % 4
    x = XA if xset == 0 else XB
    for k in range(1 if 'one' in ABL else 4):
        # ('indep', timing only: the four FMAs of a chain into four different registers -- no dependent chain)
        dst = ACC - 2 * k if 'indep' in ABL else ACC
        e("v_fmac_f64_dpp %s, %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf"
          % (vp(dst), vp(rec_w(i)), vp(x + 2 * k), 4 * q + k))


def load_next_first_half():
    """chains 0..NHALF-1 of the NEXT step (their registers are free) and its chain count"""
    load_descriptors(0, NHALF, DSTRIDE)
    if 'nosm' in ABL and _in_loop[0]:
        e("s_mov_b32 s%d, s%d" % (S_NCH_NEXT, S_NCH))
        return
    e("s_load_dword s%d, %s, 0x%x" % (S_NCH_NEXT, sp(S_DP), DSTRIDE + 4 * D_NCH))


def step_advance(last_label):
    """bookkeeping at the end of a step (in front of the last chain's FMAs / of an early exit)"""
    e("s_sub_u32 s%d, s%d, 1" % (S_NSTEP, S_NSTEP))
    e("s_cmp_eq_u32 s%d, 0" % S_NSTEP)
    br("s_cbranch_scc1", last_label)
    e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))           # record 0 of the next step
    e("s_add_u32 s%d, s%d, %d" % (S_WP, S_WP, WSTRIDE))
    e("s_addc_u32 s%d, s%d, 0" % (S_WP + 1, S_WP + 1))
    if 'dfix' not in ABL:     # (timing only: every step reads the descriptor lines of steps 0 / 1 -- scalar-cache hits)
        e("s_add_u32 s%d, s%d, %d" % (S_DP, S_DP, DSTRIDE))
        e("s_addc_u32 s%d, s%d, 0" % (S_DP + 1, S_DP + 1))


def block_one(r):
    """chain at walk position r; ONE copy of the chain loop and one row set: a chain whose successor opens a cell
    reads the new rows into the same registers right behind its own FMAs and waits for them (no cover for the LDS round
    trip, but no out-of-line block, no taken far branches, half the code: 1-2.5 % faster than a two-copy version)"""
    i, q = r // 4, r % 4
    lab("B%d_0" % r)
    if r in CHECKPOINTS and 'noexit' not in ABL:
        e("s_cmp_le_u32 s%d, %d" % (S_NCH, r))      # every chain of the pass is done: leave the step
        br("s_cbranch_scc1", "X%d" % r)
    if r % 8 == 0:
        # (the VGPR index applies to vector ALU destinations only: a build with s_set_gpr_idx_idx 0 in front of this
        # load gives the same bits and is 0.8 % slower)
        request_pair(r // 8 + AHEAD)
    if r == FORCE_WAIT:
        e("s_waitcnt lgkmcnt(0)")               # descriptors of chains NHALF.. (requested at the step's start)
    if r == NHALF:
        load_next_first_half()
    if r < NCHAIN - 1:
        rn = r + 1
        if rn % 8 == 0:
            e("s_waitcnt vmcnt(%d)" % (AHEAD - 1))     # record pair of chain rn
        select(r)
        fma4(r, 0)
        if 'nobr' in ABL:     # timing only: no new-cell test at all
            pass
        elif 'ool' in ABL:    # experiment: the new-cell block out of line, the common path falls through
            br("s_cbranch_scc1", "NC%d" % r)
            _ool.append(r)
        else:
            br("s_cbranch_scc0", "B%d_0" % rn)          # the next chain stays in the cell
            idx0()
            addresses(rn, 0)
            reads(0)
            lgkm0()
    else:
        step_advance("LAST_0")
        select(r)
        fma4(r, 0)
        lab("STEPEND")
        e("s_waitcnt lgkmcnt(0)")                    # the next step's first half and its chain count
        e("s_mov_b32 s%d, s%d" % (S_NCH, S_NCH_NEXT))
        idx0()
        addresses(0, 0)
        e("s_barrier")
        reads(0)
        lgkm0()
        load_descriptors(NHALF, NPOS - NHALF, 0)
        br("s_branch", "B0_0")


_ool = []


def new_cell_blocks():
    for r in _ool:
        lab("NC%d" % r)
        idx0()
        addresses(r + 1, 0)
        reads(0)
        lgkm0()
        br("s_branch", "B%d_0" % (r + 1))
    del _ool[:]


def exit_stubs():
    """early exits of the chain walk: issue what the skipped blocks would have issued -- the record pairs still to be
    requested (the ring and the vmcnt waits count them), the next step's first half when the walk has not reached
    position NHALF -- then the step's bookkeeping, and join the end of the step behind the last chain's FMAs"""
    for r in CHECKPOINTS:
        if 'noexit' in ABL:
            break
        lab("X%d" % r)
        for m in range(r, NPOS):
            if m % 8 == 0:
                request_pair(m // 8 + AHEAD)
        if r <= NHALF:
            load_next_first_half()
        step_advance("LASTX")
        br("s_branch", "STEPEND")
    lab("LASTX")                                     # the last step ended early: nothing left to add
    e("s_waitcnt lgkmcnt(0)")
    br("s_branch", "EPI")


def consumer():
    del L[:]
    lane_setup()
    e("v_lshlrev_b32 v%d, 3, v%d" % (V_RING, V_T0))
    e("v_and_b32 v%d, 15, v%d" % (V_L16, V_T0))
    e("v_lshlrev_b32 v%d, 4, v%d" % (V_L16, V_L16))    # (lane % 16) * 16: a lane's entry of a record pair
    read_params()
    for sreg, k in ((S_WP, P_WP), (S_WP + 1, P_WP + 1), (S_RB0, P_RB0), (S_NSTEP, P_NSTEP), (S_DP, P_DP), (S_DP + 1, P_DP + 1)):
        readlane(sreg, k)
    e("s_nop 4")
    e("v_add_u32 v%d, s%d, v%d" % (V_RING, S_RB0, V_RING))
    e("v_mov_b32 v%d, 0x200" % V_C512)
    load_descriptors(0, NHALF, 0)
    e("s_load_dword s%d, %s, 0x%x" % (S_NCH, sp(S_DP), 4 * D_NCH))
    load_descriptors(NHALF, NPOS - NHALF, 0)
    for r in range(AHEAD):
        request_pair(r)
    for j in range(NCHAIN + 1):                        # (+ the scratch accumulator of the pads)
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j))
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j + 1))
    e("s_mov_b32 s%d, 0" % S_ZERO)
    e("s_barrier")                                     # rows of steps 0..2 in LDS
    # VGPR index mode (DST_REL) for the whole loop: v_fmac_f64_dpp v[ACC + M0[7:0]]; everything else runs with M0[7:0] = 0
    e("s_set_gpr_idx_on s%d, 0x8" % S_ZERO)
    e("s_waitcnt vmcnt(%d) lgkmcnt(0)" % (AHEAD - 1))  # record 0, the descriptors of step 0
    addresses(0, 0)
    reads(0)
    e("s_waitcnt lgkmcnt(0)")                          # rows of position 0
    _in_loop[0] = True
    for r in range(NPOS):
        block_one(r)
    _in_loop[0] = False
    lab("LAST_0")
    select(NPOS - 1)
    fma4(NPOS - 1, 0)
    e("s_waitcnt lgkmcnt(0)")                          # the descriptor load ahead must not land in the epilogue's registers
    br("s_branch", "EPI")
    _in_loop[0] = True
    new_cell_blocks()
    exit_stubs()
    _in_loop[0] = False
    epilogue(XA, XB, ACC, NCHAIN, True)
    return list(L)


# =============================================================================== loader
LV_DMA, LV_HI, LV_OFF, LV_T0, LV_PAR = 1, 2, 4, 3, 46
LS_LT, LS_GROW, LS_ROWB, LS_RB0, LS_BUFB, LS_RBREQ, LS_NSTEP, LS_CNT = 4, 6, 9, 10, 11, 12, 13, 14
LS_MP, LS_MS = 20, 22   # exec masks of a row pair / of a single row
LS_G0, LS_G1, LS_G2, LS_IV, LS_NVAR = 24, 26, 28, 30, 31   # library bases of the slip variables, variable of the step
LS_TAB = 32      # [32:79] request line of a step: count, first row of the patch, requests
LS_GSTEP = 80    # [80:81] library base of the step's variable + the patch's first row
assert LS_TAB + LTABDW <= LS_GSTEP and LS_GSTEP + 1 <= S_LAST and LREQ + 2 == LTABDW
# request: rowA | (rowB - rowA) << 16 | slotA << 24, rowA relative to the patch's first row; rowB - rowA = 0: a single row;
# the two rows of a pair land in adjacent LDS slots: lanes 0-31 move rowA, lanes 32-63 rowB


def vm_wait_tree(tag, lo, hi, reg):
    """s_waitcnt vmcnt(reg) for lo <= reg <= hi (the count is an immediate: binary decision tree)"""
    if lo == hi:
        e("s_waitcnt vmcnt(%d)" % lo)
        br("s_branch", "VWD_%s" % tag)
        return
    mid = (lo + hi + 1) // 2
    e("s_cmp_lt_u32 s%d, %d" % (reg, mid))
    br("s_cbranch_scc0", "VT_%s_%d_%d" % (tag, mid, hi))
    vm_wait_tree(tag, lo, mid - 1, reg)
    lab("VT_%s_%d_%d" % (tag, mid, hi))
    vm_wait_tree(tag, mid, hi, reg)


def issue_requests(tag, nth):
    """row requests of the step whose line is in s[32:79] -> ring buffer at LS_RBREQ"""
    e("s_mul_hi_u32 s%d, s%d, s%d" % (T1, LS_TAB + 1, LS_ROWB))
    e("s_mul_i32 s%d, s%d, s%d" % (T0, LS_TAB + 1, LS_ROWB))
    e("s_add_u32 s%d, s%d, s%d" % (LS_GSTEP, LS_GROW, T0))
    e("s_addc_u32 s%d, s%d, s%d" % (LS_GSTEP + 1, LS_GROW + 1, T1))
    for k in range(LREQ):
        ent = LS_TAB + 2 + k
        e("s_cmp_le_u32 s%d, %d" % (LS_TAB, k))
        br("s_cbranch_scc1", "RQD_%s" % tag)
        e("s_and_b32 s%d, s%d, 0xffff" % (T0, ent))            # rowA
        e("s_bfe_u32 s%d, s%d, 0x80010" % (T1, ent))           # rowB - rowA
        e("s_mul_i32 s%d, s%d, s%d" % (T0, T0, LS_ROWB))
        e("s_mul_i32 s%d, s%d, s%d" % (T1, T1, LS_ROWB))
        e("s_add_u32 s%d, s%d, s%d" % (T2, LS_GSTEP, T0))
        e("s_addc_u32 s%d, s%d, 0" % (T3, LS_GSTEP + 1))
        # (the lane offsets are written under the pair mask: a single-row request in the middle of a list must not
        # leave the upper half's offsets stale)
        e("s_mov_b64 exec, %s" % sp(LS_MP))
        e("v_mad_u32_u24 v%d, v%d, s%d, v%d" % (LV_OFF, LV_HI, T1, LV_DMA))
        e("s_lshr_b32 s%d, s%d, 24" % (T0, ent))               # LDS slot of rowA
        e("s_lshl_b32 s%d, s%d, 9" % (T0, T0))
        e("s_cmp_eq_u32 s%d, 0" % T1)
        e("s_cselect_b64 exec, %s, exec" % sp(LS_MS))
        e("s_add_u32 m0, s%d, s%d" % (T0, LS_RBREQ))
        e("s_nop 0")
        if 'nodma' not in ABL:
            e("global_load_lds_dwordx4 v%d, %s%s" % (LV_OFF, sp(T2), " nt" if nth else ""))
    lab("RQD_%s" % tag)
    e("s_mov_b64 exec, -1")
    # the next step: the next slip variable's library (steps cycle through the variables)
    e("s_add_u32 s%d, s%d, 1" % (LS_IV, LS_IV))
    e("s_cmp_lt_u32 s%d, s%d" % (LS_IV, LS_NVAR))
    e("s_cselect_b32 s%d, s%d, 0" % (LS_IV, LS_IV))
    e("s_mov_b64 %s, %s" % (sp(LS_GROW), sp(LS_G0)))
    e("s_cmp_eq_u32 s%d, 1" % LS_IV)
    e("s_cselect_b64 %s, %s, %s" % (sp(LS_GROW), sp(LS_G1), sp(LS_GROW)))
    e("s_cmp_eq_u32 s%d, 2" % LS_IV)
    e("s_cselect_b64 %s, %s, %s" % (sp(LS_GROW), sp(LS_G2), sp(LS_GROW)))
    e("s_add_u32 s%d, s%d, s%d" % (LS_RBREQ, LS_RBREQ, LS_BUFB))
    e("s_mul_i32 s%d, s%d, 3" % (T0, LS_BUFB))
    e("s_add_u32 s%d, s%d, s%d" % (T0, T0, LS_RB0))
    e("s_cmp_lt_u32 s%d, s%d" % (LS_RBREQ, T0))
    e("s_cselect_b32 s%d, s%d, s%d" % (LS_RBREQ, LS_RBREQ, LS_RB0))


def load_table():
    for k in range(LTABDW // 16):
        e("s_load_dwordx16 %s, %s, 0x%x" % (sp(LS_TAB + 16 * k, 16), sp(LS_LT), 64 * k))
    e("s_add_u32 s%d, s%d, %d" % (LS_LT, LS_LT, NLOAD * LTAB))
    e("s_addc_u32 s%d, s%d, 0" % (LS_LT + 1, LS_LT + 1))


def loader(nth):
    del L[:]
    e("v_mbcnt_lo_u32_b32 v%d, -1, 0" % LV_T0)
    e("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (LV_T0, LV_T0))
    e("v_lshrrev_b32 v%d, 5, v%d" % (LV_HI, LV_T0))              # 0 for lanes 0-31 (rowA), 1 for lanes 32-63 (rowB)
    e("v_and_b32 v%d, 31, v%d" % (LV_DMA, LV_T0))
    e("v_lshlrev_b32 v%d, 4, v%d" % (LV_DMA, LV_DMA))             # byte offset of a lane inside a 512-byte row segment
    e("v_lshlrev_b32 v%d, 2, v%d" % (LV_PAR, LV_T0))
    e("v_add_u32 v%d, v%d, %%0" % (LV_PAR, LV_PAR))
    e("ds_read_b32 v%d, v%d" % (LV_PAR, LV_PAR))
    e("s_waitcnt lgkmcnt(0)")
    for sreg, k in ((LS_LT, PL_LT), (LS_LT + 1, PL_LT + 1), (LS_GROW, PL_GROW), (LS_GROW + 1, PL_GROW + 1),
                    (LS_ROWB, PL_ROWB), (LS_RB0, PL_RB0), (LS_BUFB, PL_BUFB),
                    (LS_NSTEP, PL_NSTEP), (T0, PL_NLANES), (LS_NVAR, PL_NVAR), (LS_G1, PL_G1), (LS_G1 + 1, PL_G1 + 1),
                    (LS_G2, PL_G2), (LS_G2 + 1, PL_G2 + 1)):
        e("v_readlane_b32 s%d, v%d, %d" % (sreg, LV_PAR, k))
    e("s_nop 4")
    e("s_mov_b64 %s, %s" % (sp(LS_G0), sp(LS_GROW)))
    e("s_mov_b32 s%d, 0" % LS_IV)
    # lanes that move 16 bytes of a row segment: the first NLANES of each half (pair) / of the low half (single row)
    e("s_bfm_b64 %s, s%d, 0" % (sp(LS_MS), T0))
    e("s_lshl_b64 %s, %s, 32" % (sp(LS_MP), sp(LS_MS)))
    e("s_or_b64 %s, %s, %s" % (sp(LS_MP), sp(LS_MP), sp(LS_MS)))
    e("s_mov_b32 s%d, s%d" % (LS_RBREQ, LS_RB0))
    for i in range(3):                           # rows of steps 0, 1, 2
        load_table()
        e("s_waitcnt lgkmcnt(0)")
        issue_requests("P%d" % i, nth)
    e("s_waitcnt vmcnt(0)")
    e("s_mov_b32 s%d, 0" % LS_CNT)
    e("s_barrier")
    _in_loop[0] = True
    lab("LOOP")
    load_table()                                 # requests of step s+3
    # the rows of step s+1 (requested two steps ago) have landed when at most the requests of step
    # s+2 are still in flight
    vm_wait_tree("L", 0, LREQ, LS_CNT)
    lab("VWD_L")
    e("s_barrier")
    e("s_sub_u32 s%d, s%d, 1" % (LS_NSTEP, LS_NSTEP))
    e("s_cmp_eq_u32 s%d, 0" % LS_NSTEP)
    br("s_cbranch_scc1", "LEND")
    e("s_waitcnt lgkmcnt(0)")
    e("s_mov_b32 s%d, s%d" % (LS_CNT, LS_TAB))
    issue_requests("L", nth)
    br("s_branch", "LOOP")
    _in_loop[0] = False
    lab("LEND")
    if 'nobar' in ABL:
        e("s_barrier")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return list(L)


def clobbers(vlast):
    c = ["v%d" % i for i in range(1, vlast + 1)]
    c += ["s%d" % i for i in range(2, S_LAST + 1)]
    c += ["vcc", "m0", "scc", "memory"]
    return c


# (variants 9-12 run WITHOUT loader wavefronts -- gfcell.hip lets them return --: the consumers alone, with and without
# their s_barrier per step = what the barrier's skew between the fourteen consumers costs)
VARIANTS = [set(), {"nofma"}, {"nox"}, {"nonew"}, {"norec"}, {"one"}, {"one", "norec"}, {"dfix"}, {"dfix", "norec"},
            {"nobar"}, {"noload"}, {"nobar", "norec"}, {"noload", "norec"},
            {"nobar", "nox"}, {"nobar", "nosm"}, {"nobar", "nox", "nosm", "norec"}, {"nobar", "nofma"},
            {"ool"}, {"nobr"}, {"indep"}, {"nobar", "nobr"}, {"nobar", "indep"}, {"nobar", "nobr", "indep"}]


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "beat_amd", "csrc", "gfruns_asm.inc")
    with open(out, "w") as f:
        f.write("// generated by tools/gen_gfruns_asm.py -- do not edit\n")
        f.write("// the wavefront programs of k_gfstack_runs (see gfcell.hip and the generator)\n")
        for name, val in (("NCHAIN", NCHAIN), ("NCONS", NCONS), ("NLOAD", NLOAD), ("LREQ", LREQ), ("LTABDW", LTABDW),
                          ("LTAB", LTAB), ("NVGPR", V_LAST + 1), ("SCRATCH", SCRATCH)):
            f.write("#define GC_%s %d\n" % (name, val))
        for name, val in (("WP", P_WP), ("RB0", P_RB0), ("NSTEP", P_NSTEP), ("DP", P_DP),
                          ("OUT", P_OUT), ("CTN", P_CTN), ("MODE", P_MODE), ("DATA", P_DATA), ("W", P_W),
                          ("CID", P_CID), ("PART", P_PART), ("PCS", P_PCS), ("NVALID", P_NVALID), ("TRB", P_TRB), ("WLDS", P_WLDS)):
            f.write("#define GC_P_%s %d\n" % (name, val))
        for name, val in (("LT", PL_LT), ("GROW", PL_GROW), ("ROWB", PL_ROWB), ("RB0", PL_RB0),
                          ("BUFB", PL_BUFB), ("NSTEP", PL_NSTEP), ("NLANES", PL_NLANES), ("NVAR", PL_NVAR),
                          ("G1", PL_G1), ("G2", PL_G2)):
            f.write("#define GC_PL_%s %d\n" % (name, val))
        for name, val in (("PAIR", PAIR), ("WSTRIDE", WSTRIDE), ("NHALF", NHALF), ("DLINE", DLINE), ("DHALF", DHALF),
                          ("D_BASE", D_BASE), ("D_NCH", D_NCH)):
            f.write("#define GR_%s %d\n" % (name, val))
        variants = VARIANTS if os.environ.get("GR_ABLATIONS") else VARIANTS[:1]
        f.write("#define GR_NVARIANT %d\n" % len(variants))
        cl = ", ".join('"%s"' % c for c in clobbers(V_LAST))
        for vi, abl in enumerate(variants):
            ABL.clear()
            ABL.update(abl)
            f.write("#define GR_CONSUMER_%d(PARAM_VGPR) asm volatile( \\\n" % vi)
            for line in consumer():
                f.write('    "%s\\n\\t" \\\n' % line)
            f.write('    : : "v"(PARAM_VGPR) : %s)\n' % cl)
        ABL.clear()
        cl = ", ".join('"%s"' % c for c in clobbers(LV_PAR))
        for nth in (0, 1):
            f.write("#define GC_LOADER_%d(PARAM_VGPR) asm volatile( \\\n" % nth)
            for line in loader(nth):
                f.write('    "%s\\n\\t" \\\n' % line)
            f.write('    : : "v"(PARAM_VGPR) : %s)\n' % cl)
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
