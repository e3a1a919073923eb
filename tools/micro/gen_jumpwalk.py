#!/usr/bin/env python
"""gen_jumpwalk.py -- writes tools/micro/jumpwalk.hip: gfx950 probe (round 6) of the consumer walk of k_gfstack_runs
(tools/gen_gfruns_asm.py) WITHOUT a conditional branch behind every chain.  Consumers only (no loaders, no records), the
cell pattern is static but the program treats it as data, as the kernel would:
  A   the shipped walk: per position  s_add_u32 m0 + FMAs + s_cbranch (skips the new-cell block)
  N   no branch, no new-cell block at all: the floor of the position code
  J   K copies of the straight-line walk; copy j carries a new-cell block + COMPUTED jump (s_setpc_b64, offset from the
      position's descriptor) behind every position e = j (mod K).  A cell that ends at e is walked in copy e mod K: nothing but
      s_add_u32 m0 + FMAs per chain, one indirect jump per cell.  Cells are at most K chains long (longer ones are cut).
  P   J with the rows of a cell requested ONE CELL AHEAD into a second row set (2 K copies: the row set is static per copy)
  Q   A with the same look-ahead (two copies of the loop by row set; the new-cell block ends in a branch to the other copy)
  I   N with the FMAs of two positions interleaved (s_add_u32 m0 in front of every FMA): two independent dependency chains
nf = FMAs per position: 4 = multilinear (four corner rows per cell), 1 = nearest neighbour (a "cell" is one row).
python gen_jumpwalk.py && hipcc --offload-arch=gfx950 -O2 jumpwalk.hip -o jumpwalk
"""
import os

RUNS = [1, 5, 3, 7]          # cell lengths, cyclic over 16 positions (4 chains per cell)
K = 8


def cell_ends(npos):
    ends, r = [], 0
    i = 0
    while r < npos:
        r += RUNS[i % len(RUNS)]
        ends.append(min(r, npos) - 1)
        i += 1
    return ends


def prog(mode, npos, nf, nsteps_reg="s2", blk="full"):
    L = []
    e = L.append
    ACC = 48
    W, X, AD, RING, C = 4, 8, 40, 44, 45
    ends = cell_ends(npos)
    isend = [r in ends for r in range(npos)]
    e("s_mov_b32 s3, 0")
    for k in range(16):
        slot = (k * 7) % npos
        e("s_mov_b32 s%d, 0x%x" % (32 + 2 * k, 0x4000 | slot | ((1 if isend[k] else 0) << 31)))
        e("s_mov_b32 s%d, 0x%x" % (33 + 2 * k, ((k * 5) % 48) | (((k * 11) % 48) << 16)))
    if blk == "desync":
        # wave w walks the pattern rotated by 4 w positions: the wavefronts of a SIMD do not reach their blocks together
        for k in range(16):
            for w in range(1, 4):
                kk = (k + 4 * w) % 16
                e("s_cmp_eq_u32 s1, %d" % w)
                e("s_cselect_b32 s%d, 0x%x, s%d" % (32 + 2 * k, 0x4000 | ((k * 7) % npos) | ((1 if isend[kk] else 0) << 31), 32 + 2 * k))
    if mode in "JP":
        e("s_getpc_b64 s[90:91]")
        e("BASE_%=:")
        # jump targets as the table kernel would hand them over: one per cell (+ the step end)
        for i, en in enumerate(ends):
            start = 0 if i == 0 else ends[i - 1] + 1
            if mode == "P":
                e("s_mov_b32 s%d, Q%d_%d_%d_%%=-BASE_%%=" % (64 + i, i % 2, en % K, start))
            else:
                e("s_mov_b32 s%d, P%d_%d_%%=-BASE_%%=" % (64 + i, en % K, start))
        e("s_mov_b32 s%d, END_%%=-BASE_%%=" % (64 + len(ends)))
    e("s_set_gpr_idx_on s3, 0x8")
    e("STEP_%=:")

    def reads(d):
        e("s_set_gpr_idx_idx s3")
        e("v_mad_u32_u16 v%d, s%d, v%d, v%d" % (AD, d + 1, C, RING))
        if nf == 4:
            e("v_mad_u32_u16 v%d, s%d, v%d, v%d op_sel:[1,0,0,0]" % (AD + 1, d + 1, C, RING))
            for q, (a, off) in enumerate(((AD + 1, 512), (AD + 1, 0), (AD, 512), (AD, 0))):
                e("ds_read_b64 v[%d:%d], v%d offset:%d" % (X + 2 * q, X + 2 * q + 1, a, off))
        else:
            e("ds_read_b64 v[%d:%d], v%d" % (X, X + 1, AD))

    def fmas(r):
        for q in range(nf):
            e("v_fmac_f64_dpp v[%d:%d], v[%d:%d], v[%d:%d] row_newbcast:%d row_mask:0xf bank_mask:0xf"
              % (ACC, ACC + 1, W, W + 1, X + 2 * q, X + 2 * q + 1, (nf * r + q) % 16))

    def jump(i):          # to the start of cell i (i == len(ends): the step end)
        t = 64 + i
        e("s_bfe_u32 s92, s%d, 0x100000" % t)
        e("s_add_u32 s92, s92, s90")
        e("s_addc_u32 s93, s91, 0")
        e("s_waitcnt lgkmcnt(0)")
        e("s_setpc_b64 s[92:93]")

    X2 = 24

    def reads_to(d, xs):
        e("s_set_gpr_idx_idx s3")
        e("v_mad_u32_u16 v%d, s%d, v%d, v%d" % (AD, d + 1, C, RING))
        if nf == 4:
            e("v_mad_u32_u16 v%d, s%d, v%d, v%d op_sel:[1,0,0,0]" % (AD + 1, d + 1, C, RING))
            for q, (a, off) in enumerate(((AD + 1, 512), (AD + 1, 0), (AD, 512), (AD, 0))):
                e("ds_read_b64 v[%d:%d], v%d offset:%d" % (xs + 2 * q, xs + 2 * q + 1, a, off))
        else:
            e("ds_read_b64 v[%d:%d], v%d" % (xs, xs + 1, AD))

    def fmas_from(r, xs):
        for q in range(nf):
            e("v_fmac_f64_dpp v[%d:%d], v[%d:%d], v[%d:%d] row_newbcast:%d row_mask:0xf bank_mask:0xf"
              % (ACC, ACC + 1, W, W + 1, xs + 2 * q, xs + 2 * q + 1, (nf * r + q) % 16))

    if mode == "P":
        # cell i reads row set i % 2; the block behind cell i-1 waits for the rows of cell i, requests those of cell i+1
        # into ITS OWN set (set (i-1) % 2 = (i+1) % 2, its FMAs are issued) and jumps to cell i in the other set's copy
        reads_to(32, X)                     # cell 0 -> set 0
        e("s_waitcnt lgkmcnt(0)")
        if len(ends) > 1:
            reads_to(32, X2)                # cell 1 -> set 1
        e("s_bfe_u32 s92, s64, 0x100000")
        e("s_add_u32 s92, s92, s90")
        e("s_addc_u32 s93, s91, 0")
        e("s_setpc_b64 s[92:93]")
        for st in range(2):
            xs = (X, X2)[st]
            for j in range(K):
                for r in range(npos):
                    d = 32 + 2 * (r % 16)
                    e("Q%d_%d_%d_%%=:" % (st, j, r))
                    e("s_add_u32 m0, s%d, s%d" % (d, d))
                    fmas_from(r, xs)
                    if r % K == j or r == npos - 1:
                        i = ends.index(r) + 1 if r in ends else len(ends)      # the cell the jump goes to
                        e("s_waitcnt lgkmcnt(0)")
                        if i + 1 < len(ends):
                            reads_to(d, xs)
                        e("s_bfe_u32 s92, s%d, 0x100000" % (64 + i))
                        e("s_add_u32 s92, s92, s90")
                        e("s_addc_u32 s93, s91, 0")
                        e("s_setpc_b64 s[92:93]")
        e("END_%=:")
    elif mode == "Q":
        reads_to(32, X)
        e("s_waitcnt lgkmcnt(0)")
        reads_to(32, X2)
        for st in range(2):
            xs = (X, X2)[st]
            for r in range(npos):
                d = 32 + 2 * (r % 16)
                e("R%d_%d_%%=:" % (st, r))
                e("s_add_u32 m0, s%d, s%d" % (d, d))
                fmas_from(r, xs)
                if r == npos - 1:
                    e("s_waitcnt lgkmcnt(0)")
                    e("s_branch END_%=")
                else:
                    e("s_cbranch_scc0 R%d_%d_%%=" % (st, r + 1))
                    e("s_waitcnt lgkmcnt(0)")
                    reads_to(d, xs)
                    e("s_branch R%d_%d_%%=" % (1 - st, r + 1))
        e("END_%=:")
    elif mode == "I":
        for r in range(0, npos - 1, 2):
            d0, d1 = 32 + 2 * (r % 16), 32 + 2 * ((r + 1) % 16)
            for q in range(nf):
                for rr, dd in ((r, d0), (r + 1, d1)):
                    e("s_add_u32 m0, s%d, s%d" % (dd, dd))
                    e("v_fmac_f64_dpp v[%d:%d], v[%d:%d], v[%d:%d] row_newbcast:%d row_mask:0xf bank_mask:0xf"
                      % (ACC, ACC + 1, W, W + 1, X + 2 * q, X + 2 * q + 1, (nf * rr + q) % 16))
        r = npos - 1
        e("s_add_u32 m0, s%d, s%d" % (32 + 2 * (r % 16), 32 + 2 * (r % 16)))
        fmas(r)
    elif mode == "J":
        reads(32)
        jump(0)
        for j in range(K):
            for r in range(npos):
                d = 32 + 2 * (r % 16)
                e("P%d_%d_%%=:" % (j, r))
                e("s_add_u32 m0, s%d, s%d" % (d, d))
                fmas(r)
                if r % K == j or r == npos - 1:
                    # (the kernel's table hands the target over in the descriptor of position r: static register per position)
                    i = ends.index(r) + 1 if r in ends else len(ends)
                    if i < len(ends):
                        reads(d)
                    jump(i)
        e("END_%=:")
    elif mode == "A" and blk in ("emptyinv", "inv", "invq"):
        # the branch is TAKEN only when the next chain opens a cell (10 of 37); inv: the blocks out of line, a branch back;
        # invq: two copies by row set, the block (look-ahead reads as Q) ends in the branch into the other copy
        if blk == "invq":
            reads_to(32, X)
            e("s_waitcnt lgkmcnt(0)")
            reads_to(32, X2)
        ncopy = 2 if blk == "invq" else 1
        for st in range(ncopy):
            xs = (X, X2)[st]
            for r in range(npos):
                d = 32 + 2 * (r % 16)
                e("R%d_%d_%%=:" % (st, r))
                e("s_add_u32 m0, s%d, s%d" % (d, d))
                fmas_from(r, xs)
                if r < npos - 1:
                    e("s_cbranch_scc1 %s%d_%d_%%=" % ("R" if blk == "emptyinv" else "B", st, r + 1 if blk == "emptyinv" else r))
            e("s_branch END_%=")
        if blk != "emptyinv":
            for st in range(ncopy):
                xs = (X, X2)[st]
                for r in range(npos - 1):
                    d = 32 + 2 * (r % 16)
                    e("B%d_%d_%%=:" % (st, r))
                    if blk == "invq":
                        e("s_waitcnt lgkmcnt(0)")
                        reads_to(d, xs)
                        e("s_branch R%d_%d_%%=" % (1 - st, r + 1))
                    else:
                        reads_to(d, xs)
                        e("s_waitcnt lgkmcnt(0)")
                        e("s_branch R%d_%d_%%=" % (st, r + 1))
        e("END_%=:")
        if blk == "invq":
            e("s_waitcnt lgkmcnt(0)")
    else:
        for r in range(npos):
            d = 32 + 2 * (r % 16)
            e("s_add_u32 m0, s%d, s%d" % (d, d))
            fmas(r)
            if mode == "A":
                # blk: timing experiments on the new-cell block (what in it costs)
                e("s_cbranch_scc0 P%d_%%=" % (r + 1))
                if blk in ("full", "desync"):
                    reads(d)
                    e("s_waitcnt lgkmcnt(0)")
                elif blk == "nowait":
                    reads(d)
                elif blk == "noread":
                    e("s_set_gpr_idx_idx s3")
                    e("v_mad_u32_u16 v%d, s%d, v%d, v%d" % (AD, d + 1, C, RING))
                    if nf == 4:
                        e("v_mad_u32_u16 v%d, s%d, v%d, v%d op_sel:[1,0,0,0]" % (AD + 1, d + 1, C, RING))
                elif blk == "noidx":
                    L2 = []
                    e2 = e
                    n0 = len(L)
                    reads(d)
                    del L[n0]          # the s_set_gpr_idx_idx
                    e("s_waitcnt lgkmcnt(0)")
                elif blk == "onlyidx":
                    e("s_set_gpr_idx_idx s3")
                elif blk == "nomad":
                    e("s_set_gpr_idx_idx s3")
                    if nf == 4:
                        for q, (a, off) in enumerate(((RING, 512), (RING, 0), (RING, 1536), (RING, 1024))):
                            e("ds_read_b64 v[%d:%d], v%d offset:%d" % (X + 2 * q, X + 2 * q + 1, a, off))
                    else:
                        e("ds_read_b64 v[%d:%d], v%d" % (X, X + 1, RING))
                    e("s_waitcnt lgkmcnt(0)")
                e("P%d_%%=:" % (r + 1))
    e("s_barrier")
    e("s_sub_u32 %s, %s, 1" % (nsteps_reg, nsteps_reg))
    e("s_cmp_eq_u32 %s, 0" % nsteps_reg)
    e("s_cbranch_scc0 STEP_%=")
    e("s_set_gpr_idx_off")
    return L


def kernel(name, mode, npos, nf, nthreads, blk="full"):
    out = []
    out.append("__global__ void __launch_bounds__(%d) %s(double *out, int nsteps)\n{" % (nthreads, name))
    out.append("    extern __shared__ __attribute__((aligned(16))) double rows[];")
    out.append("    const int lane = threadIdx.x & 63;")
    out.append("    for (int i = threadIdx.x; i < 104 * 64; i += blockDim.x) rows[i] = 1e-3 * (i % 977);")
    out.append("    __syncthreads();")
    out.append("    const unsigned ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)rows + lane * 8;")
    out.append("    asm volatile(")
    pre = ["v_mov_b32 v44, %0", "v_mov_b32 v45, 0x200", "s_mov_b32 s2, %1", "s_mov_b32 s1, %2",
           "v_mov_b32 v4, 0x3a83126f", "v_mov_b32 v5, 0x3f50624d"]
    for v in range(8, 40):
        pre.append("v_mov_b32 v%d, 0" % v)
    for v in range(48, 128):
        pre.append("v_mov_b32 v%d, 0" % v)
    for line in pre + prog(mode, npos, nf, blk=blk):
        out.append('        "%s\\n\\t"' % line)
    cl = ['"v%d"' % i for i in range(1, 128)] + ['"s%d"' % i for i in range(1, 96)] + ['"vcc"', '"m0"', '"scc"', '"memory"']
    out.append('        : : "v"(ring), "s"(nsteps), "s"(__builtin_amdgcn_readfirstlane((threadIdx.x >> 6) >> 2) & 3) : %s);' % ", ".join(cl))
    out.append("    if (out && nsteps < 0) out[threadIdx.x] = rows[lane];")
    out.append("}")
    return "\n".join(out)


CONFIGS = [("kA4_16", "A", 37, 4, 16), ("kN4_16", "N", 37, 4, 16), ("kJ4_16", "J", 37, 4, 16), ("kP4_16", "P", 37, 4, 16), ("kQ4_16", "Q", 37, 4, 16),
           ("kA4_desync", "A", 37, 4, 16, "desync"), ("kA4_nowait", "A", 37, 4, 16, "nowait"), ("kA4_noread", "A", 37, 4, 16, "noread"),
           ("kA4_noidx", "A", 37, 4, 16, "noidx"), ("kA4_onlyidx", "A", 37, 4, 16, "onlyidx"), ("kA4_nomad", "A", 37, 4, 16, "nomad"),
           ("kA4_empty", "A", 37, 4, 16, "empty"), ("kA4_emptyinv", "A", 37, 4, 16, "emptyinv"), ("kA4_inv", "A", 37, 4, 16, "inv"),
           ("kA4_invq", "A", 37, 4, 16, "invq"), ("kA4_inv14", "A", 37, 4, 14, "inv"), ("kA4_invq14", "A", 37, 4, 14, "invq"), ("kA4_14", "A", 37, 4, 14),
           ("kP4_8", "P", 37, 4, 8), ("kA4_8", "A", 37, 4, 8),
           ("kA1_16", "A", 37, 1, 16), ("kN1_16", "N", 37, 1, 16), ("kA1_desync", "A", 37, 1, 16, "desync"), ("kA1_empty", "A", 37, 1, 16, "empty"),
           ("kA1_onlyidx", "A", 37, 1, 16, "onlyidx"), ("kA1_emptyinv", "A", 37, 1, 16, "emptyinv"), ("kA1_inv", "A", 37, 1, 16, "inv"), ("kA1_invq", "A", 37, 1, 16, "invq")]


def main():
    src = ["// generated by tools/micro/gen_jumpwalk.py -- see there", "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstdint>"]
    for cfg in CONFIGS:
        name, mode, npos, nf, nw = cfg[:5]
        src.append(kernel(name, mode, npos, nf, nw * 64, *cfg[5:]))
    src.append("int main()\n{\n    double *out; hipMalloc(&out, 1 << 20);\n    const int nsteps = 4000;")
    for cfg in CONFIGS:
        name, mode, npos, nf, nw = cfg[:5]
        src.append("""    {
        hipFuncSetAttribute((const void *)%(name)s, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(%(name)s, dim3(256), dim3(%(nth)d), 150 * 1024, 0, out, nsteps);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double ns = ms * 1e6 / nsteps;
        printf("%(name)-8s mode %(mode)s %(nf)d FMA/position %(nw)2d waves x %(npos)d positions: %%.3f ms, %%.0f ns per step, %%.2f ns per position and SIMD "
               "(FMAs alone at 2.4 GHz: %%.2f)  %%s\\n", ms, ns, ns / (%(npos)d * %(nw)d / 4.0), %(nf)d * 4 / 2.4, hipGetErrorString(hipGetLastError()));
    }""" % dict(name=name, mode=mode, npos=npos, nth=nw * 64, nf=nf, nw=nw))
    src.append("    return 0;\n}")
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jumpwalk.hip")
    open(p, "w").write("\n".join(src) + "\n")
    print("wrote", p)


if __name__ == "__main__":
    main()
