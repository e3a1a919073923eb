#!/usr/bin/env python
"""
gen_gfcell_asm.py -- writes beat_amd/csrc/gfcell_asm.inc: the two wavefront programs of
k_gfstack_cell (beat_amd/csrc/gfcell.hip), the multilinear Green's-function stacking kernel for
gfx950 whose accumulators are addressed through the VGPR index register (s_set_gpr_idx_*).

Reference arithmetic: beat/ffi/base.py:607-709 (multilinear branch :663-704) -- per (chain, target,
sample): acc = fma(G[row_k], w_k, acc) for the four corner rows k of the chain's (duration,
start-time) cell, patches in ascending order.  Same operations in the same order as k_gfstack
(gfstack.hip): bitwise equal.

Mapping: workgroup = (518-chain group, target, 64-sample tile) = 14 consumer
wavefronts + 2 loader wavefronts.
  consumer: 37 chains, lane <-> sample, accumulator of chain j = v[ACC+2j : ACC+2j+1].  Per patch
      the chains are visited cell by cell (batches of up to four chains of one cell): the four
      rows of the cell are read ONCE from LDS into registers (contiguous 512-byte reads, no bank
      conflicts) and applied to every chain of the batch by
          v_fmac_f64_dpp acc[M0], w, x_k  row_newbcast:(4q+k)
      -- the accumulator is selected through M0 (s_set_gpr_idx_on, DST_REL), the weight is lane
      4q+k of every 16-lane row of a VGPR pair that one coalesced load filled with the sixteen
      weights of the batch.  The batch records (sixteen weights + a descriptor: accumulator
      indices, chain count, LDS offsets of the next batch's rows) come four at a time ("quad", 1 KB)
      by ONE vector load per quad, bounced through a wavefront-private LDS buffer and re-read
      record by record with 16-lane-replicated ds_reads one record ahead: the vector memory path
      handles a wave instruction in ~16 cycles whatever it moves (two loads per record made the
      texture addresser the bottleneck of an earlier version: TA busy 86 %), scalar loads return
      out of order (one block of look-ahead per wavefront: the first version was bound by that
      latency); LDS operations are cheap to issue and complete in order.
  loader:   LDS-DMA (global_load_lds_dwordx4) of every distinct row segment of the group into a
      ring of three LDS row buffers, two patches ahead, from a per-step request table.
One s_barrier per patch for all sixteen wavefronts.

The programs are generated because they are register-allocated by hand (the accumulators must be a
contiguous physical VGPR range) and unrolled over the buffer rings.

    python tools/gen_gfcell_asm.py        # rewrites beat_amd/csrc/gfcell_asm.inc
"""
import os

NCHAIN = 37            # chains per consumer wavefront
NCONS, NLOAD = 14, 2   # consumer / loader wavefronts
QREC = 256             # bytes per batch record inside a quad: 16 weights at +0, 16 descriptor dwords at +128
QUAD = 4 * QREC        # four records = one vector load (64 lanes x 16 B)
NQMAX = 12             # quads per (wavefront, step) in the record table
WSTRIDE = NQMAX * QUAD
NQMIN = 3              # a step has at least this many quads (empty records are appended): the quad a
                       # wavefront requests is two ahead of the one it works on
BOUNCE = 2 * QUAD      # bytes of LDS per consumer wavefront: two quads
LPAIR = 31             # row-pair requests per loader and step
LTAB = 32 * 4          # bytes per (loader, step) in the request table
# descriptor dwords of a batch record (lane k of every 16-lane row of the AUX register holds dword k)
A_XN = 0               # [0:4)  LDS byte offsets of the four rows of the NEXT batch
A_ACC01, A_CF, A_ACC23 = 4, 5, 6   # M0 words (0x8000 | 2j) of chains 0,1 / flags / chains 2,3
A_XO = 8               # [8:12) LDS byte offsets of this batch's own rows (used for the first batch of a step)
CF_LAST, CF_CROSS = 3, 4   # A_CF: chain count in bits 0-2; last record of the step; (first record of a quad) the
                           # quad requested now is the last of its step
M0_IDX0 = 0x8000       # VGPR index mode: DST_REL, index 0

# ---------------------------------------------------------------- consumer registers
V_IN = 0         # "%0": LDS address of the wavefront's parameter block (allocated by the compiler)
V_RING = 1       # lane*8 + LDS address of the row ring (epilogue: lane*8)
V_WB = 2         # bounce buffer + (lane % 16) * 8: a lane's weight of a record
V_AB = 3         # bounce buffer + (lane % 16) * 4: a lane's descriptor dword of a record
V_T0 = 4
V_PAR = 5        # parameter dwords (lane k = dword k)
V_AD = 6         # [6:9] LDS addresses of the four rows
V_L16 = 10       # lane*16: a lane's 16 bytes of a quad in memory
V_BW = 11        # bounce buffer + lane*16
V_S = 12         # [12:15] the quad in flight
AUXA, AUXB = 16, 17
WA, WB_ = 18, 20
XA, XB = 22, 30  # row register sets: 4 pairs each
V_D = XA         # [+0:1] data of the tile (epilogue: the row registers are free)
V_T1 = XA + 2
V_T2 = XA + 4
ACC = 38
V_LAST = ACC + 2 * NCHAIN - 1

S_NSTEP = 2
S_WP, S_WN = 4, 6
S_CF, S_A01, S_A23 = 8, 9, 10
T0, T1, T2, T3 = 16, 17, 18, 19   # (T2:T3 is an aligned pair: addresses)
S_RB0, S_BNC = 15, 14
S_LAST = 95      # user SGPRs stop here: VCC, FLAT_SCRATCH and XNACK_MASK take the top six

# parameter block of a consumer wavefront (dwords)
P_WP, P_RB0, P_NSTEP, P_BNC = 0, 4, 5, 6
P_DP = 8         # k_gfstack_runs: the wavefront's chain descriptors (scalar loads)
P_OUT, P_CTN, P_MODE, P_DATA, P_W, P_CID, P_PART, P_PCS, P_NVALID, P_TRB = 16, 18, 19, 20, 22, 24, 26, 28, 29, 30
# parameter block of a loader wavefront
PL_LT, PL_GROW, PL_DSRB, PL_ROWB, PL_RB0, PL_BUFB, PL_NSTEP, PL_NLANES = 0, 2, 4, 5, 6, 7, 8, 9
PL_NVAR, PL_G1, PL_G2 = 10, 12, 14   # slip variables (steps cycle through their libraries patch by patch), bases 2 and 3

L = []
ABL = set()   # timing experiments: 'nofma', 'nox', 'nobar', 'now' (results are wrong with any of them)


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


# =============================================================================== consumer
def x_prefetch(aux, first, xn):
    """rows of the next batch (descriptor dwords A_XN..) or of this batch (A_XO..) -> register set xn"""
    for k in range(4):
        e("v_add_u32_dpp v%d, v%d, v%d row_newbcast:%d row_mask:0xf bank_mask:0xf" % (V_AD + k, aux, V_RING, first + k))
    for k in range(4):
        e("ds_read_b64 %s, v%d" % (vp(xn + 2 * k), V_AD + k))


def rec_regs(v):
    """registers of record position v of the 8-record cycle: (weights, descriptor, rows)"""
    return (WA, AUXA, XA) if v % 2 == 0 else (WB_, AUXB, XB)


def rec_off(v):
    return (v // 4) * QUAD + (v % 4) * QREC


def read_record(v):
    """weights and descriptor of record position v: bounce buffer -> registers (16-lane replicated)"""
    w, aux, _ = rec_regs(v)
    e("ds_read_b64 %s, v%d offset:%d" % (vp(w), V_WB, rec_off(v)))
    e("ds_read_b32 v%d, v%d offset:%d" % (aux, V_AB, rec_off(v) + 128))


def request_quad():
    e("global_load_dwordx4 v[%d:%d], v%d, %s" % (V_S, V_S + 3, V_L16, sp(S_WP)))


def batch(v):
    w, aux, xc = rec_regs(v)
    _, _, xn = rec_regs(v + 1)
    nxt = (v + 1) % 8
    qp, r = v // 4, v % 4
    lab("BATCH_%d" % v)
    readlane(S_CF, A_CF, aux)
    readlane(S_A01, A_ACC01, aux)
    readlane(S_A23, A_ACC23, aux)
    if r == 0:
        # a new quad: the one after it has landed -> the other half of the bounce buffer; request
        # the quad after that
        e("s_waitcnt vmcnt(0)")
        e("ds_write_b128 v%d, v[%d:%d] offset:%d" % (V_BW, V_S, V_S + 3, (1 - qp) * QUAD))
        request_quad()
        e("s_bitcmp1_b32 s%d, %d" % (S_CF, CF_CROSS))
        br("s_cbranch_scc1", "CROSS_%d" % v)
        e("s_add_u32 s%d, s%d, %d" % (S_WP, S_WP, QUAD))
        e("s_addc_u32 s%d, s%d, 0" % (S_WP + 1, S_WP + 1))
        lab("ADV_%d" % v)
    # rows, weights and descriptor of the next record (the last record of a step names slot 0 of
    # the ring: read, not used)
    x_prefetch(aux, A_XN, xn)
    read_record(nxt)
    # chains 3..0 of the batch: entry by the count (bits 0-2 of the descriptor)
    e("s_bitcmp1_b32 s%d, 2" % S_CF)
    br("s_cbranch_scc1", "C4_%d" % v)
    e("s_bitcmp1_b32 s%d, 1" % S_CF)
    br("s_cbranch_scc0", "LOW_%d" % v)
    e("s_bitcmp1_b32 s%d, 0" % S_CF)
    br("s_cbranch_scc1", "C3_%d" % v)
    br("s_branch", "C2_%d" % v)
    lab("LOW_%d" % v)
    e("s_bitcmp1_b32 s%d, 0" % S_CF)
    br("s_cbranch_scc1", "C1_%d" % v)
    br("s_branch", "TAIL_%d" % v)
    for q in (3, 2, 1, 0):
        lab("C%d_%d" % (q + 1, v))
        src = S_A23 if q >= 2 else S_A01
        if q % 2:
            e("s_lshr_b32 m0, s%d, 16" % src)
        else:
            e("s_and_b32 m0, s%d, 0xffff" % src)
        for k in range(4):
            e("v_fmac_f64_dpp %s, %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf"
              % (vp(ACC), vp(w), vp(xc + 2 * k), 4 * q + k))
    lab("TAIL_%d" % v)
    e("s_mov_b32 m0, 0x%x" % M0_IDX0)
    e("s_waitcnt lgkmcnt(0)")                          # the next record and its rows are in registers
    if r == 3:
        e("s_bitcmp1_b32 s%d, %d" % (S_CF, CF_LAST))
        br("s_cbranch_scc1", "STEPEND_%d" % v)
    if nxt == 0:
        br("s_branch", "BATCH_0")


def out_of_line(v):
    w, aux, xc = rec_regs(v)
    naux, nx = rec_regs(v + 1)[1], rec_regs(v + 1)[2]
    nxt = (v + 1) % 8
    if v % 4 == 0:
        lab("CROSS_%d" % v)
        e("s_mov_b64 %s, %s" % (sp(S_WP), sp(S_WN)))
        e("s_add_u32 s%d, s%d, %d" % (S_WN, S_WN, WSTRIDE))
        e("s_addc_u32 s%d, s%d, 0" % (S_WN + 1, S_WN + 1))
        br("s_branch", "ADV_%d" % v)
    if v % 4 == 3:
        lab("STEPEND_%d" % v)
        e("s_sub_u32 s%d, s%d, 1" % (S_NSTEP, S_NSTEP))
        e("s_cmp_eq_u32 s%d, 0" % S_NSTEP)
        br("s_cbranch_scc1", "EPI")
        e("s_barrier")                                 # rows of the next step published by the loaders
        x_prefetch(naux, A_XO, nx)                     # the first record of a step reads its own rows
        e("s_waitcnt lgkmcnt(0)")
        br("s_branch", "BATCH_%d" % nxt)


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
    T4, T5 = 20, 21
    for sreg, k in ((S_OUT, P_OUT), (S_OUT + 1, P_OUT + 1), (S_CTN, P_CTN), (S_MODE, P_MODE),
                    (S_DATA, P_DATA), (S_DATA + 1, P_DATA + 1), (S_W, P_W), (S_W + 1, P_W + 1),
                    (S_CID, P_CID), (S_CID + 1, P_CID + 1), (S_PART, P_PART), (S_PART + 1, P_PART + 1),
                    (S_PCS, P_PCS), (S_NVAL, P_NVALID), (S_TRB, P_TRB)):
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
    lab("END")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")


def consumer():
    del L[:]
    lane_setup()
    e("v_lshlrev_b32 v%d, 3, v%d" % (V_RING, V_T0))
    e("v_lshlrev_b32 v%d, 4, v%d" % (V_L16, V_T0))
    e("v_and_b32 v%d, 15, v%d" % (V_AB, V_T0))
    e("v_lshlrev_b32 v%d, 3, v%d" % (V_WB, V_AB))
    e("v_lshlrev_b32 v%d, 2, v%d" % (V_AB, V_AB))
    read_params()
    for sreg, k in ((S_WP, P_WP), (S_WP + 1, P_WP + 1), (S_RB0, P_RB0), (S_NSTEP, P_NSTEP), (S_BNC, P_BNC)):
        readlane(sreg, k)
    e("s_nop 4")
    e("v_add_u32 v%d, s%d, v%d" % (V_RING, S_RB0, V_RING))
    e("v_add_u32 v%d, s%d, v%d" % (V_WB, S_BNC, V_WB))
    e("v_add_u32 v%d, s%d, v%d" % (V_AB, S_BNC, V_AB))
    e("v_add_u32 v%d, s%d, v%d" % (V_BW, S_BNC, V_L16))
    for j in range(NCHAIN):
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j))
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j + 1))
    e("s_add_u32 s%d, s%d, %d" % (S_WN, S_WP, WSTRIDE))
    e("s_addc_u32 s%d, s%d, 0" % (S_WN + 1, S_WP + 1))
    # quad 0 -> first half of the bounce buffer, quad 1 in flight, record 0 in registers
    request_quad()
    e("s_add_u32 s%d, s%d, %d" % (S_WP, S_WP, QUAD))
    e("s_addc_u32 s%d, s%d, 0" % (S_WP + 1, S_WP + 1))
    e("s_waitcnt vmcnt(0)")
    e("ds_write_b128 v%d, v[%d:%d]" % (V_BW, V_S, V_S + 3))
    request_quad()
    e("s_add_u32 s%d, s%d, %d" % (S_WP, S_WP, QUAD))
    e("s_addc_u32 s%d, s%d, 0" % (S_WP + 1, S_WP + 1))
    read_record(0)
    e("s_barrier")                                     # rows of steps 0..2 in LDS
    # VGPR index mode stays on for the whole loop: M0 = 0x8000 | 2 * chain slot selects the accumulator
    # of the v_fmac_f64_dpp; every other vector instruction runs with index 0
    e("s_mov_b32 s%d, 0" % T0)
    e("s_set_gpr_idx_on s%d, 0x8" % T0)
    e("s_waitcnt lgkmcnt(0)")
    x_prefetch(AUXA, A_XO, XA)
    e("s_waitcnt lgkmcnt(0)")
    _in_loop[0] = True
    for v in range(8):
        batch(v)
    for v in range(8):
        out_of_line(v)
    _in_loop[0] = False
    epilogue(XA, XB, ACC, NCHAIN, True)
    return list(L)


# =============================================================================== loader
LV_DMA, LV_HI, LV_OFF, LV_T0, LV_PAR = 1, 2, 4, 3, 46
LS_LT, LS_GROW, LS_DSRB, LS_ROWB, LS_RB0, LS_BUFB, LS_RBREQ, LS_NSTEP, LS_CNT = 4, 6, 8, 9, 10, 11, 12, 13, 14
LS_MP, LS_MS = 20, 22   # exec masks of a row pair / of a single row
LS_G0, LS_G1, LS_G2, LS_IV, LS_NVAR, LS_ROWOFF = 24, 26, 28, 30, 31, 15   # library bases, variable of the step, rows of the patch
LS_TAB = 32      # [32:63] request table of a step: count, requests
# request: rowA | (rowB - rowA) << 8 | slotA << 16 | single << 24  (rows relative to the step's first row;
# the two rows of a pair land in adjacent LDS slots: lanes 0-31 move rowA, lanes 32-63 rowB)


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
    """row requests of the step whose table is in s[32:63] -> ring buffer at LS_RBREQ"""
    for k in range(LPAIR):
        ent = LS_TAB + 1 + k
        e("s_cmp_le_u32 s%d, %d" % (LS_TAB, k))
        br("s_cbranch_scc1", "RQD_%s" % tag)
        e("s_and_b32 s%d, s%d, 0xff" % (T0, ent))
        e("s_bfe_u32 s%d, s%d, 0x80008" % (T1, ent))           # rowB - rowA
        e("s_mul_i32 s%d, s%d, s%d" % (T0, T0, LS_ROWB))
        e("s_mul_i32 s%d, s%d, s%d" % (T1, T1, LS_ROWB))
        e("s_add_u32 s%d, s%d, s%d" % (T2, LS_GROW, T0))
        e("s_addc_u32 s%d, s%d, 0" % (T3, LS_GROW + 1))
        # (the lane offsets are written under the pair mask: a single-row request in the middle of a list -- the
        # dense layout of k_gfstack_ml has them -- must not leave the upper half's offsets stale)
        e("s_mov_b64 exec, %s" % sp(LS_MP))
        e("v_mad_u32_u24 v%d, v%d, s%d, v%d" % (LV_OFF, LV_HI, T1, LV_DMA))
        e("s_bfe_u32 s%d, s%d, 0x80010" % (T0, ent))           # LDS slot of rowA
        e("s_lshl_b32 s%d, s%d, 9" % (T0, T0))
        e("s_bitcmp1_b32 s%d, 24" % ent)
        e("s_cselect_b64 exec, %s, exec" % sp(LS_MS))
        e("s_add_u32 m0, s%d, s%d" % (T0, LS_RBREQ))
        e("s_nop 0")
        if 'nodma' not in ABL:
            e("global_load_lds_dwordx4 v%d, %s%s" % (LV_OFF, sp(T2), " nt" if nth else ""))
    lab("RQD_%s" % tag)
    e("s_mov_b64 exec, -1")
    # the next step: the next slip variable's library at the same patch, or the first one at the next patch
    e("s_add_u32 s%d, s%d, 1" % (LS_IV, LS_IV))
    e("s_cmp_lt_u32 s%d, s%d" % (LS_IV, LS_NVAR))
    br("s_cbranch_scc1", "SAMEP_%s" % tag)
    e("s_mov_b32 s%d, 0" % LS_IV)
    e("s_add_u32 s%d, s%d, s%d" % (LS_ROWOFF, LS_ROWOFF, LS_DSRB))
    lab("SAMEP_%s" % tag)
    e("s_mov_b64 %s, %s" % (sp(LS_GROW), sp(LS_G0)))
    e("s_cmp_eq_u32 s%d, 1" % LS_IV)
    e("s_cselect_b64 %s, %s, %s" % (sp(LS_GROW), sp(LS_G1), sp(LS_GROW)))
    e("s_cmp_eq_u32 s%d, 2" % LS_IV)
    e("s_cselect_b64 %s, %s, %s" % (sp(LS_GROW), sp(LS_G2), sp(LS_GROW)))
    e("s_add_u32 s%d, s%d, s%d" % (LS_GROW, LS_GROW, LS_ROWOFF))
    e("s_addc_u32 s%d, s%d, 0" % (LS_GROW + 1, LS_GROW + 1))
    e("s_add_u32 s%d, s%d, s%d" % (LS_RBREQ, LS_RBREQ, LS_BUFB))
    e("s_mul_i32 s%d, s%d, 3" % (T0, LS_BUFB))
    e("s_add_u32 s%d, s%d, s%d" % (T0, T0, LS_RB0))
    e("s_cmp_lt_u32 s%d, s%d" % (LS_RBREQ, T0))
    e("s_cselect_b32 s%d, s%d, s%d" % (LS_RBREQ, LS_RBREQ, LS_RB0))


def load_table():
    for k in range(2):
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
                    (LS_DSRB, PL_DSRB), (LS_ROWB, PL_ROWB), (LS_RB0, PL_RB0), (LS_BUFB, PL_BUFB),
                    (LS_NSTEP, PL_NSTEP), (T0, PL_NLANES), (LS_NVAR, PL_NVAR), (LS_G1, PL_G1), (LS_G1 + 1, PL_G1 + 1),
                    (LS_G2, PL_G2), (LS_G2 + 1, PL_G2 + 1)):
        e("v_readlane_b32 s%d, v%d, %d" % (sreg, LV_PAR, k))
    e("s_nop 4")
    e("s_mov_b64 %s, %s" % (sp(LS_G0), sp(LS_GROW)))
    e("s_mov_b32 s%d, 0" % LS_IV)
    e("s_mov_b32 s%d, 0" % LS_ROWOFF)
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
    vm_wait_tree("L", 0, LPAIR, LS_CNT)
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


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "beat_amd", "csrc",
                       "gfcell_asm.inc")
    with open(out, "w") as f:
        f.write("// generated by tools/gen_gfcell_asm.py -- do not edit\n")
        f.write("// the wavefront programs of k_gfstack_cell (see gfcell.hip and the generator)\n")
        for name, val in (("NCHAIN", NCHAIN), ("NCONS", NCONS), ("NLOAD", NLOAD), ("QREC", QREC), ("QUAD", QUAD),
                          ("NQMAX", NQMAX), ("WSTRIDE", WSTRIDE), ("NQMIN", NQMIN), ("BOUNCE", BOUNCE),
                          ("CF_LAST", CF_LAST), ("CF_CROSS", CF_CROSS),
                          ("A_XN", A_XN), ("A_ACC01", A_ACC01), ("A_CF", A_CF), ("A_ACC23", A_ACC23), ("A_XO", A_XO),
                          ("LPAIR", LPAIR), ("LTAB", LTAB), ("NVGPR", V_LAST + 1)):
            f.write("#define GC_%s %d\n" % (name, val))
        for name, val in (("WP", P_WP), ("RB0", P_RB0), ("NSTEP", P_NSTEP), ("BNC", P_BNC), ("DP", P_DP),
                          ("OUT", P_OUT), ("CTN", P_CTN), ("MODE", P_MODE), ("DATA", P_DATA), ("W", P_W),
                          ("CID", P_CID), ("PART", P_PART), ("PCS", P_PCS), ("NVALID", P_NVALID), ("TRB", P_TRB)):
            f.write("#define GC_P_%s %d\n" % (name, val))
        for name, val in (("LT", PL_LT), ("GROW", PL_GROW), ("DSRB", PL_DSRB), ("ROWB", PL_ROWB), ("RB0", PL_RB0),
                          ("BUFB", PL_BUFB), ("NSTEP", PL_NSTEP), ("NLANES", PL_NLANES), ("NVAR", PL_NVAR),
                          ("G1", PL_G1), ("G2", PL_G2)):
            f.write("#define GC_PL_%s %d\n" % (name, val))
        variants = [set()] + ([{"nobar"}, {"nodma"}] if os.environ.get("GC_ABLATIONS") else [])
        f.write("#define GC_NVARIANT %d\n" % len(variants))
        for vi, abl in enumerate(variants):
            ABL.clear()
            ABL.update(abl)
            cl = ", ".join('"%s"' % c for c in clobbers(V_LAST))
            f.write("#define GC_CONSUMER_%d(PARAM_VGPR) asm volatile( \\\n" % vi)
            for line in consumer():
                f.write('    "%s\\n\\t" \\\n' % line)
            f.write('    : : "v"(PARAM_VGPR) : %s)\n' % cl)
            cl = ", ".join('"%s"' % c for c in clobbers(LV_PAR))
            for nth in (0, 1):
                f.write("#define GC_LOADER_%d_%d(PARAM_VGPR) asm volatile( \\\n" % (vi, nth))
                for line in loader(nth):
                    f.write('    "%s\\n\\t" \\\n' % line)
                f.write('    : : "v"(PARAM_VGPR) : %s)\n' % cl)
        ABL.clear()
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
