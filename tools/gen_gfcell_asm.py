#!/usr/bin/env python
"""
gen_gfcell_asm.py -- writes beat_amd/csrc/gfcell_asm.inc: the wavefront program of
k_gfstack_cell (beat_amd/csrc/gfcell.hip), the multilinear Green's-function stacking kernel for
gfx950 whose accumulators are addressed through the VGPR index register (s_set_gpr_idx_*).

Reference arithmetic: beat/ffi/base.py:607-709 (multilinear branch :663-704) -- per (chain, target,
sample): acc = fma(G[row_k], w_k, acc) for the four corner rows k of the chain's (duration,
start-time) cell, patches in ascending order.  Same operations in the same order as k_gfstack
(gfstack.hip): bitwise equal.

Mapping: workgroup = (512-chain group, target, 64-sample tile) = 16 wavefronts;
a wavefront owns 32 chains, lane <-> sample, accumulator of chain j = v[ACC+2j : ACC+2j+1].
The chains of a wavefront are processed cell by cell: the four rows of a cell are read ONCE from
LDS into registers (contiguous 512-byte reads, no bank conflicts) and applied to every chain of
the cell with the chain's four weights as SGPR operands; the accumulator is selected by
M0 (DST_REL | SRC2_REL).  Everything a wavefront does is a command stream written by
k_gc_tables: 144-byte blocks = the weights of up to four chains of one cell + the LDS offsets of
the next block's rows + accumulator indices + flags, read with scalar loads one block ahead.

The program is generated because it is fully register-allocated by hand (the accumulators must be
a contiguous physical VGPR range) and unrolled over two SGPR/VGPR buffer sets.

    python tools/gen_gfcell_asm.py        # rewrites beat_amd/csrc/gfcell_asm.inc
"""
import os

# ---------------------------------------------------------------- register map
V_RING = 32      # lane*8 + LDS address of the row ring
V_DMA = 33       # lane*16: global byte offset of a lane inside a 512-byte row segment
V_T0 = 34        # temporaries
V_PAR = 35       # parameter dwords (lane k = dword k), kept for the epilogue
V_D = 36         # [36:37] data of the tile (epilogue)
V_T1 = 38        # [38:39] temporary pair
XA, XB = 40, 48  # row register sets: 4 pairs each
V_AD = 56        # [56:59] LDS addresses of the four rows
V_T2 = 60        # [60:61] temporary pair (epilogue)
V_LANE8 = 61     # (epilogue only, aliases V_T2+1: set where needed)
ACC = 62         # [62:125] 32 accumulators
NCHAIN = 32
V_LAST = ACC + 2 * NCHAIN - 1   # 125

# persistent scalar state (user SGPRs stop at s95: VCC, FLAT_SCRATCH and XNACK_MASK take the top six)
S_NSTEP = 3      # steps left
S_ST = 4         # [4:5]   stream pointer (next block to load)
S_HD = 6         # [6:7]   header pointer (next header to load)
T0, T1, T2, T3, T4, T5 = 8, 9, 10, 11, 12, 13
AW, AI = 16, 48  # set A: weights s[16:47], info s[48:51]
BW, BI = 52, 84  # set B: weights s[52:83], info s[84:87]
S_HDR = 88       # [88:95] step header: count, up to 7 requests (relrow | slot << 16)
# The state of the row requests is only needed between two steps.  It lives in lanes of V_PAR and is
# brought into the SGPRs of the buffer set that has just been consumed (offsets from its base):
Q_GROW, Q_DSRB, Q_ROWB, Q_RB0, Q_BUFB, Q_RBREQ, Q_CNT = 0, 2, 3, 4, 5, 6, 7
BLOCK = 144      # bytes per stream block
HDR_STRIDE = 16 * 32  # bytes per step in the header array (16 wavefronts x 8 dwords)

# parameter block (dwords) written by the C++ prologue into LDS, one block per wavefront
P_ST, P_HD, P_GROW, P_DSRB, P_ROWB, P_RB0, P_BUFB, P_NSTEP, P_NLANES, P_RBREQ, P_CNT = 0, 2, 4, 6, 7, 8, 9, 10, 11, 12, 13
P_OUT, P_CTN, P_MODE, P_DATA, P_W, P_CID, P_PART, P_PCS, P_NVALID, P_TRB = 16, 18, 19, 20, 22, 24, 26, 28, 29, 30

L = []


def e(s):
    L.append(s)


def lab(name):
    e("%s_%%=:" % name)


def br(op, name):
    e("%s %s_%%=" % (op, name))


def vp(r):
    return "v[%d:%d]" % (r, r + 1)


def sp(r, n=2):
    return "s[%d:%d]" % (r, r + n - 1)


def readlane(sreg, k):
    e("v_readlane_b32 s%d, v%d, %d" % (sreg, V_PAR, k))


def req_state_load(q):
    """request state: lanes of V_PAR -> s[q .. q+7]"""
    for off, k in ((Q_GROW, P_GROW), (Q_GROW + 1, P_GROW + 1), (Q_DSRB, P_DSRB), (Q_ROWB, P_ROWB),
                   (Q_RB0, P_RB0), (Q_BUFB, P_BUFB), (Q_RBREQ, P_RBREQ), (Q_CNT, P_CNT)):
        readlane(q + off, k)
    e("s_nop 4")


def req_state_store(q):
    for off, k in ((Q_GROW, P_GROW), (Q_GROW + 1, P_GROW + 1), (Q_RBREQ, P_RBREQ), (Q_CNT, P_CNT)):
        e("v_writelane_b32 v%d, s%d, %d" % (V_PAR, q + off, k))


def issue_requests(tag, nth, q):
    """row requests of the step whose header is in s[88:95] -> ring buffer at RBREQ; state in s[q..]"""
    e("s_mov_b64 exec, vcc")
    for k in range(7):
        e("s_cmp_le_u32 s%d, %d" % (S_HDR, k))
        br("s_cbranch_scc1", "RQD_%s" % tag)
        e("s_and_b32 s%d, s%d, 0xffff" % (T0, S_HDR + 1 + k))
        e("s_lshr_b32 s%d, s%d, 16" % (T1, S_HDR + 1 + k))
        e("s_mul_i32 s%d, s%d, s%d" % (T0, T0, q + Q_ROWB))
        e("s_add_u32 s%d, s%d, s%d" % (T2, q + Q_GROW, T0))
        e("s_addc_u32 s%d, s%d, 0" % (T3, q + Q_GROW + 1))
        e("s_lshl_b32 s%d, s%d, 9" % (T1, T1))
        e("s_add_u32 m0, s%d, s%d" % (T1, q + Q_RBREQ))
        e("s_nop 0")
        e("global_load_lds_dwordx4 v%d, %s%s" % (V_DMA, sp(T2), " nt" if nth else ""))
    lab("RQD_%s" % tag)
    e("s_mov_b64 exec, -1")
    # next step: library base, ring buffer
    e("s_add_u32 s%d, s%d, s%d" % (q + Q_GROW, q + Q_GROW, q + Q_DSRB))
    e("s_addc_u32 s%d, s%d, 0" % (q + Q_GROW + 1, q + Q_GROW + 1))
    e("s_add_u32 s%d, s%d, s%d" % (q + Q_RBREQ, q + Q_RBREQ, q + Q_BUFB))
    e("s_mul_i32 s%d, s%d, 3" % (T0, q + Q_BUFB))
    e("s_add_u32 s%d, s%d, s%d" % (T0, T0, q + Q_RB0))
    e("s_cmp_lt_u32 s%d, s%d" % (q + Q_RBREQ, T0))
    e("s_cselect_b32 s%d, s%d, s%d" % (q + Q_RBREQ, q + Q_RBREQ, q + Q_RB0))


def x_prefetch(info, xn):
    """rows of the NEXT block (LDS offsets in 8-byte units in info dwords 0-1) -> register set xn"""
    e("s_and_b32 s%d, s%d, 0xffff" % (T0, info))
    e("s_lshr_b32 s%d, s%d, 16" % (T1, info))
    e("s_and_b32 s%d, s%d, 0xffff" % (T2, info + 1))
    e("s_lshr_b32 s%d, s%d, 16" % (T3, info + 1))
    for k, t in enumerate((T0, T1, T2, T3)):
        e("v_lshl_add_u32 v%d, s%d, 3, v%d" % (V_AD + k, t, V_RING))
    for k in range(4):
        e("ds_read_b64 %s, v%d" % (vp(xn + 2 * k), V_AD + k))


def batch(tag, w, info, x, nw, ninfo, nx, ntag, nth):
    lab("BATCH_%s" % tag)
    e("s_bitcmp1_b32 s%d, 3" % (info + 3))
    br("s_cbranch_scc1", "LASTPRE_%s" % tag)
    x_prefetch(info, nx)
    lab("LOADS_%s" % tag)
    e("s_load_dwordx16 %s, %s, 0x0" % (sp(nw, 16), sp(S_ST)))
    e("s_load_dwordx16 %s, %s, 0x40" % (sp(nw + 16, 16), sp(S_ST)))
    e("s_load_dwordx4 %s, %s, 0x80" % (sp(ninfo, 4), sp(S_ST)))
    e("s_lshr_b32 s%d, s%d, 8" % (T0, info + 3))
    e("s_add_u32 s%d, s%d, %d" % (T0, T0, BLOCK))
    e("s_add_u32 s%d, s%d, s%d" % (S_ST, S_ST, T0))
    e("s_addc_u32 s%d, s%d, 0" % (S_ST + 1, S_ST + 1))
    e("s_and_b32 s%d, s%d, 7" % (T4, info + 3))
    for k in range(4):
        e("s_cmp_le_u32 s%d, %d" % (T4, k))
        br("s_cbranch_scc1", "DONE_%s" % tag)
        if k == 0:
            e("s_set_gpr_idx_on s%d, 0xc" % (info + 2))
        else:
            e("s_lshr_b32 s%d, s%d, %d" % (T5, info + 2, 8 * k))
            e("s_set_gpr_idx_on s%d, 0xc" % T5)
        for r in range(4):
            e("v_fma_f64 %s, %s, %s, %s" % (vp(ACC), vp(x + 2 * r), sp(w + 8 * k + 2 * r), vp(ACC)))
    lab("DONE_%s" % tag)
    e("s_set_gpr_idx_off")
    e("s_waitcnt lgkmcnt(0)")
    e("s_bitcmp1_b32 s%d, 3" % (info + 3))
    br("s_cbranch_scc0", "BATCH_%s" % ntag)
    # ---- end of a step
    e("s_sub_u32 s%d, s%d, 1" % (S_NSTEP, S_NSTEP))
    e("s_cmp_eq_u32 s%d, 0" % S_NSTEP)
    br("s_cbranch_scc1", "EPI")
    # the rows of the next step (requested two steps ago) have landed when at most the requests
    # of the step after it are still in flight
    req_state_load(w)
    e("s_lshr_b32 s%d, s%d, 8" % (T0, w + Q_CNT))
    for k in range(7):
        e("s_cmp_eq_u32 s%d, %d" % (T0, k))
        br("s_cbranch_scc1", "VW%d_%s" % (k, tag))
    for k in range(7, -1, -1):
        lab("VW%d_%s" % (k, tag))
        e("s_waitcnt vmcnt(%d)" % k)
        if k:
            br("s_branch", "VWD_%s" % tag)
    lab("VWD_%s" % tag)
    e("s_barrier")
    # outstanding counts: (s+1) <- (s+2), (s+2) <- the requests issued now
    e("s_lshr_b32 s%d, s%d, 8" % (T0, w + Q_CNT))
    e("s_lshl_b32 s%d, s%d, 8" % (T1, S_HDR))
    e("s_or_b32 s%d, s%d, s%d" % (w + Q_CNT, T0, T1))
    issue_requests("E" + tag, nth, w)
    req_state_store(w)
    br("s_branch", "BATCH_%s" % ntag)
    lab("LASTPRE_%s" % tag)
    e("s_load_dwordx8 %s, %s, 0x0" % (sp(S_HDR, 8), sp(S_HD)))
    e("s_add_u32 s%d, s%d, %d" % (S_HD, S_HD, HDR_STRIDE))
    e("s_addc_u32 s%d, s%d, 0" % (S_HD + 1, S_HD + 1))
    br("s_branch", "LOADS_%s" % tag)


def program(nth):
    del L[:]
    # ---------------------------------------------------------------- prologue
    e("v_mbcnt_lo_u32_b32 v%d, -1, 0" % V_T0)
    e("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (V_T0, V_T0))
    e("v_lshlrev_b32 v%d, 4, v%d" % (V_DMA, V_T0))
    e("v_lshlrev_b32 v%d, 3, v%d" % (V_RING, V_T0))
    e("v_lshlrev_b32 v%d, 2, v%d" % (V_T0, V_T0))
    e("v_add_u32 v%d, v%d, %%0" % (V_T0, V_T0))
    e("ds_read_b32 v%d, v%d" % (V_PAR, V_T0))
    e("s_waitcnt lgkmcnt(0)")
    for sreg, k in ((S_ST, P_ST), (S_ST + 1, P_ST + 1), (S_HD, P_HD), (S_HD + 1, P_HD + 1),
                    (S_NSTEP, P_NSTEP), (T0, P_NLANES)):
        readlane(sreg, k)
    req_state_load(AW)
    e("s_bfm_b64 vcc, s%d, 0" % T0)              # lanes that move 16 bytes of a row segment
    e("s_mov_b32 s%d, s%d" % (AW + Q_RBREQ, AW + Q_RB0))
    e("s_mov_b32 s%d, 0" % (AW + Q_CNT))
    e("v_add_u32 v%d, s%d, v%d" % (V_RING, AW + Q_RB0, V_RING))
    for j in range(NCHAIN):
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j))
        e("v_mov_b32 v%d, 0" % (ACC + 2 * j + 1))
    # rows of steps 0, 1, 2
    for i in range(3):
        e("s_load_dwordx8 %s, %s, 0x0" % (sp(S_HDR, 8), sp(S_HD)))
        e("s_add_u32 s%d, s%d, %d" % (S_HD, S_HD, HDR_STRIDE))
        e("s_addc_u32 s%d, s%d, 0" % (S_HD + 1, S_HD + 1))
        e("s_waitcnt lgkmcnt(0)")
        issue_requests("P%d" % i, nth, AW)
    req_state_store(AW)
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    # the lead block of step 0
    e("s_load_dwordx16 %s, %s, 0x0" % (sp(AW, 16), sp(S_ST)))
    e("s_load_dwordx16 %s, %s, 0x40" % (sp(AW + 16, 16), sp(S_ST)))
    e("s_load_dwordx4 %s, %s, 0x80" % (sp(AI, 4), sp(S_ST)))
    e("s_add_u32 s%d, s%d, %d" % (S_ST, S_ST, BLOCK))
    e("s_addc_u32 s%d, s%d, 0" % (S_ST + 1, S_ST + 1))
    e("s_waitcnt lgkmcnt(0)")
    # ---------------------------------------------------------------- main loop
    batch("A", AW, AI, XA, BW, BI, XB, "B", nth)
    batch("B", BW, BI, XB, AW, AI, XA, "A", nth)
    # ---------------------------------------------------------------- epilogue
    lab("EPI")
    e("s_barrier")   # every wavefront is done with the row ring
    S_OUT, S_CTN, S_MODE, S_DATA, S_W, S_CID, S_PART, S_PCS, S_NVAL, S_TRB = 16, 18, 19, 20, 22, 24, 26, 28, 29, 30
    for sreg, k in ((S_OUT, P_OUT), (S_OUT + 1, P_OUT + 1), (S_CTN, P_CTN), (S_MODE, P_MODE),
                    (S_DATA, P_DATA), (S_DATA + 1, P_DATA + 1), (S_W, P_W), (S_W + 1, P_W + 1),
                    (S_CID, P_CID), (S_CID + 1, P_CID + 1), (S_PART, P_PART), (S_PART + 1, P_PART + 1),
                    (S_PCS, P_PCS), (S_NVAL, P_NVALID), (S_TRB, P_TRB)):
        readlane(sreg, k)
    e("s_nop 4")
    CID = 52   # s[52:83] chain ids of the 32 accumulators
    e("s_load_dwordx16 %s, %s, 0x0" % (sp(CID, 16), sp(S_CID)))
    e("s_load_dwordx16 %s, %s, 0x40" % (sp(CID + 16, 16), sp(S_CID)))
    # lane*8 and the mask of valid samples
    e("v_mbcnt_lo_u32_b32 v%d, -1, 0" % V_T0)
    e("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (V_T0, V_T0))
    e("v_lshlrev_b32 v%d, 3, v%d" % (V_RING, V_T0))          # V_RING = lane*8 from here on
    e("s_mov_b64 vcc, -1")
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
                tmp = V_T1 if (j & 1) == 0 else XA
                e("v_add_f64 %s, %s, -%s" % (vp(tmp), vp(V_D), vp(ACC + 2 * j)))   # seismic.py:1332
                e("global_store_dwordx2 v%d, %s, %s" % (V_RING, vp(tmp), sp(T2)))
            else:
                e("global_store_dwordx2 v%d, %s, %s" % (V_RING, vp(ACC + 2 * j), sp(T2)))
            lab("SK%s%d" % (tag, j))
        e("s_mov_b64 exec, -1")
        br("s_branch", "END")

    # ---- residual store
    store_loop("R", True)
    # ---- synthetics
    lab("SYN")
    store_loop("S", False)
    # ---- scalar-covariance misfit: partial[c, t, tile] = sum_i (w (d_i - syn_i))^2, i ascending
    lab("SCAL")
    V_WA, V_RA, V_C = V_AD, V_AD + 1, V_AD + 2
    Q = V_T2
    TPITCH = 65 * 8
    e("v_add_u32 v%d, s%d, v%d" % (V_WA, S_TRB, V_RING))            # write: region + lane*8
    e("v_lshrrev_b32 v%d, 3, v%d" % (V_T0, V_RING))                 # lane
    e("v_mul_u32_u24 v%d, %d, v%d" % (V_RA, TPITCH, V_T0))
    e("v_add_u32 v%d, s%d, v%d" % (V_RA, S_TRB, V_RA))              # read: region + lane*pitch
    e("v_lshlrev_b32 v%d, 2, v%d" % (V_AD + 3, V_T0))               # lane*4
    for r in range(2):
        for jj in range(16):
            j = 16 * r + jj
            tmp = V_T1 if (jj & 1) == 0 else XB
            e("v_add_f64 %s, %s, -%s" % (vp(tmp), vp(V_D), vp(ACC + 2 * j)))
            e("v_mul_f64 %s, %s, %s" % (vp(tmp), sp(S_W), vp(tmp)))     # distributions.py:128 with W = w I
            e("v_cndmask_b32 v%d, 0, v%d, vcc" % (tmp, tmp))             # samples beyond N contribute 0
            e("v_cndmask_b32 v%d, 0, v%d, vcc" % (tmp + 1, tmp + 1))
            e("ds_write_b64 v%d, %s offset:%d" % (V_WA, vp(tmp), jj * TPITCH))
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v%d, 0" % Q)
        e("v_mov_b32 v%d, 0" % (Q + 1))
        e("s_mov_b64 exec, 0xffff")
        e("global_load_dword v%d, v%d, %s offset:%d" % (V_C, V_AD + 3, sp(S_CID), 64 * r))
        for i0 in range(0, 64, 8):
            for k in range(8):
                e("ds_read_b64 %s, v%d offset:%d" % (vp(XA + 2 * k), V_RA, (i0 + k) * 8))
            e("s_waitcnt lgkmcnt(0)")
            for k in range(8):
                e("v_fma_f64 %s, %s, %s, %s" % (vp(Q), vp(XA + 2 * k), vp(XA + 2 * k), vp(Q)))
        e("s_waitcnt vmcnt(0)")
        # address of partial[(c*T + t)*ntile + tile] = PART + c * PCS
        e("v_mov_b32 v%d, s%d" % (V_T1, S_PART))
        e("v_mov_b32 v%d, s%d" % (V_T1 + 1, S_PART + 1))
        e("v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (vp(XB), sp(T2), V_C, S_PCS, vp(V_T1)))
        e("v_cmp_ne_u32 %s, -1, v%d" % (sp(T2), V_C))
        e("s_and_b64 exec, exec, %s" % sp(T2))
        e("global_store_dwordx2 %s, %s, off" % (vp(XB), vp(Q)))
        e("s_mov_b64 exec, -1")
        e("s_waitcnt vmcnt(0)")
    lab("END")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return list(L)


def clobbers():
    c = ["v%d" % i for i in range(V_RING, V_LAST + 1)]
    c += ["s%d" % i for i in range(3, 96)]
    c += ["vcc", "m0", "scc", "memory"]
    return c


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "beat_amd", "csrc",
                       "gfcell_asm.inc")
    with open(out, "w") as f:
        f.write("// generated by tools/gen_gfcell_asm.py -- do not edit\n")
        f.write("// the wavefront program of k_gfstack_cell (see gfcell.hip and the generator)\n")
        f.write("#define GC_NCHAIN %d\n#define GC_BLOCK %d\n#define GC_HDR_STRIDE %d\n" % (NCHAIN, BLOCK, HDR_STRIDE))
        f.write("#define GC_NVGPR %d\n" % (V_LAST + 1))
        for name, val in (("ST", P_ST), ("HD", P_HD), ("GROW", P_GROW), ("DSRB", P_DSRB), ("ROWB", P_ROWB),
                          ("RB0", P_RB0), ("BUFB", P_BUFB), ("NSTEP", P_NSTEP), ("NLANES", P_NLANES),
                          ("OUT", P_OUT), ("CTN", P_CTN), ("MODE", P_MODE), ("DATA", P_DATA), ("W", P_W),
                          ("CID", P_CID), ("PART", P_PART), ("PCS", P_PCS), ("NVALID", P_NVALID), ("TRB", P_TRB)):
            f.write("#define GC_P_%s %d\n" % (name, val))
        cl = ", ".join('"%s"' % c for c in clobbers())
        for nth in (0, 1):
            f.write("#define GC_PROGRAM_%d(PARAM_VGPR) asm volatile( \\\n" % nth)
            for line in program(nth):
                f.write('    "%s\\n\\t" \\\n' % line)
            f.write('    : : "v"(PARAM_VGPR) : %s)\n' % cl)
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
