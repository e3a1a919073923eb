#!/usr/bin/env python
"""gen_twosample.py -- writes tools/micro/twosample.hip: gfx950 probe (round 6) of the consumer walk of k_gfstack_runs
(tools/gen_gfruns_asm.py) in two register layouts, consumers only (no loaders, no records):
  A   one sample per lane: 4 waves per SIMD at 128 VGPRs, per position  s_add_u32 m0 + 4 dependent v_fmac_f64_dpp + branch,
      a new cell every ~4 positions = 2 v_mad + 4 ds_read_b64 + wait                      (the shipped walk)
  B   two samples per lane: 2 waves per SIMD at 256 VGPRs, per position s_add_u32 m0 + 8 v_fmac_f64_dpp (two interleaved
      dependency chains) + branch, a new cell = 2 v_mad + 4 ds_read_b128 + wait
  C   B with the next cell's rows requested BEFORE the FMAs of the position that ends a cell (second row set)
chain-samples per ns and CU for each.   python gen_twosample.py && hipcc --offload-arch=gfx950 -O2 twosample.hip -o twosample
"""
import os

RUNS = [1, 5, 3, 7]          # cell lengths, cyclic over 16 positions (4 chains per cell)
NEW = []
for n in RUNS:
    NEW += [0] * (n - 1) + [1]          # NEW[r % 16]: the position BEHIND r opens a cell


def prog(mode, npos, nsteps_reg="s2"):
    two = mode in "BC"
    L = []
    e = L.append
    ACC = 64 if two else 48
    W, X, X2, AD, RING, C = 4, 8, 24, 40, 44, 45      # weights pair, row set(s), addresses, ring address, slot pitch
    step = 4 if two else 2
    e("s_mov_b32 s3, 0")
    for k in range(16):
        slot = (k * 7) % npos
        e("s_mov_b32 s%d, 0x%x" % (32 + 2 * k, 0x4000 | (slot * step // 2) | (NEW[k] << 31)))
        e("s_mov_b32 s%d, 0x%x" % (33 + 2 * k, ((k * 5) % 48) | (((k * 11) % 48) << 16)))
    e("s_set_gpr_idx_on s3, 0x8")
    e("STEP_%=:")
    for r in range(npos):
        k = r % 16
        d = 32 + 2 * k

        def reads(xs):
            e("s_set_gpr_idx_idx s3")
            e("v_mad_u32_u16 v%d, s%d, v%d, v%d" % (AD, d + 1, C, RING))
            e("v_mad_u32_u16 v%d, s%d, v%d, v%d op_sel:[1,0,0,0]" % (AD + 1, d + 1, C, RING))
            if two:
                for q, (a, off) in enumerate(((AD + 1, 1024), (AD + 1, 0), (AD, 1024), (AD, 0))):
                    e("ds_read_b128 v[%d:%d], v%d offset:%d" % (xs + 4 * q, xs + 4 * q + 3, a, off))
            else:
                for q, (a, off) in enumerate(((AD + 1, 512), (AD + 1, 0), (AD, 512), (AD, 0))):
                    e("ds_read_b64 v[%d:%d], v%d offset:%d" % (xs + 2 * q, xs + 2 * q + 1, a, off))

        def fmas(xs):
            for q in range(4):
                if two:
                    e("v_fmac_f64_dpp v[%d:%d], v[%d:%d], v[%d:%d] row_newbcast:%d row_mask:0xf bank_mask:0xf"
                      % (ACC, ACC + 1, W, W + 1, xs + 4 * q, xs + 4 * q + 1, (4 * r + q) % 16))
                    e("v_fmac_f64_dpp v[%d:%d], v[%d:%d], v[%d:%d] row_newbcast:%d row_mask:0xf bank_mask:0xf"
                      % (ACC + 2, ACC + 3, W, W + 1, xs + 4 * q + 2, xs + 4 * q + 3, (4 * r + q) % 16))
                else:
                    e("v_fmac_f64_dpp v[%d:%d], v[%d:%d], v[%d:%d] row_newbcast:%d row_mask:0xf bank_mask:0xf"
                      % (ACC, ACC + 1, W, W + 1, xs + 2 * q, xs + 2 * q + 1, (4 * r + q) % 16))

        if mode == "C":
            # static pattern: the row set alternates per cell; a position that ends a cell requests the next cell's rows
            # into the other set in front of its own FMAs, waits behind them
            cell = sum(NEW[(q % 16)] for q in range(r)) & 1
            xs, xo = (X, X2) if cell == 0 else (X2, X)
            if NEW[k]:
                reads(xo)
            e("s_add_u32 m0, s%d, s%d" % (d, d))
            fmas(xs)
            if NEW[k]:
                e("s_waitcnt lgkmcnt(0)")
            else:
                e("s_cbranch_scc0 P%d_%%=" % (r + 1))      # (taken, as in the shipped walk; the skipped block is empty here)
                e("s_nop 0")
        else:
            e("s_add_u32 m0, s%d, s%d" % (d, d))
            fmas(X)
            e("s_cbranch_scc0 P%d_%%=" % (r + 1))
            reads(X)
            e("s_waitcnt lgkmcnt(0)")
        e("P%d_%%=:" % (r + 1))
    e("s_barrier")
    e("s_sub_u32 %s, %s, 1" % (nsteps_reg, nsteps_reg))
    e("s_cmp_eq_u32 %s, 0" % nsteps_reg)
    e("s_cbranch_scc0 STEP_%=")
    e("s_set_gpr_idx_off")
    return L


def kernel(name, mode, npos, nthreads):
    two = mode in "BC"
    vmax = 255 if two else 127
    acc0 = 64 if two else 48
    out = []
    out.append("__global__ void __launch_bounds__(%d) %s(double *out, int nsteps)\n{" % (nthreads, name))
    out.append("    extern __shared__ __attribute__((aligned(16))) double rows[];")
    out.append("    const int lane = threadIdx.x & 63;")
    out.append("    for (int i = threadIdx.x; i < 104 * 64; i += blockDim.x) rows[i] = 1e-3 * (i %% 977);" .replace("%%", "%"))
    out.append("    __syncthreads();")
    out.append("    const unsigned ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)rows + lane * %d;" % (16 if two else 8))
    out.append("    asm volatile(")
    pre = ["v_mov_b32 v44, %0", "v_mov_b32 v45, 0x%x" % (1024 if two else 512), "s_mov_b32 s2, %1",
           "v_mov_b32 v4, 0x3a83126f", "v_mov_b32 v5, 0x3f50624d"]
    for v in range(8, 40):
        pre.append("v_mov_b32 v%d, 0" % v)
    for v in range(acc0, vmax + 1):
        pre.append("v_mov_b32 v%d, 0" % v)
    for line in pre + prog(mode, npos):
        out.append('        "%s\\n\\t"' % line)
    cl = ['"v%d"' % i for i in range(1, vmax + 1)] + ['"s%d"' % i for i in range(2, 96)] + ['"vcc"', '"m0"', '"scc"', '"memory"']
    out.append('        : : "v"(ring), "s"(nsteps) : %s);' % ", ".join(cl))
    out.append("    if (out && nsteps < 0) out[threadIdx.x] = rows[lane];")
    out.append("}")
    return "\n".join(out)


CONFIGS = [("kA14", "A", 37, 14 * 64, 64), ("kA16", "A", 37, 16 * 64, 64),
           ("kB8", "B", 52, 8 * 64, 128), ("kB7", "B", 52, 7 * 64, 128), ("kB8_44", "B", 44, 8 * 64, 128),
           ("kC8", "C", 48, 8 * 64, 128), ("kC7", "C", 48, 7 * 64, 128), ("kB4", "B", 52, 4 * 64, 128)]


def main():
    src = ["// generated by tools/micro/gen_twosample.py -- see there", "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstdint>"]
    for name, mode, npos, nth, ns in CONFIGS:
        src.append(kernel(name, mode, npos, nth))
    src.append("int main()\n{\n    double *out; hipMalloc(&out, 1 << 20);\n    const int nsteps = 4000;")
    for name, mode, npos, nth, ns in CONFIGS:
        src.append("""    {
        hipFuncSetAttribute((const void *)%(name)s, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(%(name)s, dim3(256), dim3(%(nth)d), 150 * 1024, 0, out, nsteps);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double cs = (double)(%(nth)d / 64) * %(npos)d * %(ns)d * nsteps;     // chain-samples per CU
        printf("%(name)-8s mode %(mode)s %(nw)2d waves x %(npos)d positions x %(ns)d samples: %%.3f ms, %%.0f ns per step, %%.2f chain-samples per ns and CU "
               "(FP64 peak at 2.4 GHz: 38.4)  %%s\\n", ms, ms * 1e6 / nsteps, cs / (ms * 1e6), hipGetErrorString(hipGetLastError()));
    }""" % dict(name=name, mode=mode, npos=npos, nth=nth, ns=ns, nw=nth // 64))
    src.append("    return 0;\n}")
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "twosample.hip")
    open(p, "w").write("\n".join(src) + "\n")
    print("wrote", p)


if __name__ == "__main__":
    main()
