#!/usr/bin/env python
"""
bench.py -- SMC chain-steps/s of the FFI seismic hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2], SURVEY.md 8(d) "config 3"): one 20x20-patch subfault
(P=400, 1 km), T=64 targets, N=4096 samples, library (64,400,3,25,4096) float64 = 62.9 GB
resident in HBM, N(0,1) values generated on device (synthetic), covariance sigma^2 I
(``--covariance toeplitz`` for the dense-W variant), nearest-neighbour interpolation
(``--interp multilinear`` for the 4-row blend).

One "step" = one batched Metropolis.astep (reference beat/sampler/metropolis.py:276-422)
for ``--chains`` chains per GPU: proposal, prior-box check, forward model
(fast sweep -> start times -> GF stacking -> residual -> sum logp), tempered MH accept.
All inputs (library, data, weights, chain states, proposal rows, uniforms) are resident in
HBM when the timed region starts.

    python bench.py                        # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Chains are sharded over ranks (weak scaling: fixed chains per GPU); there is no collective
inside a step -- the SMC stage transition (all-gather of end points, once per ~100-400
steps) is timed separately and reported as ``stage_transition_ms``.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes_per_chain_step(spec, nvar=1):
    """gathered rows + index/slip tables + the data read of the fused
    residual (replaces the synthetics write of the unfused form, same size)."""
    T, P, N = spec.T, spec.P, spec.N
    rows = 4 if spec.interpolation == "multilinear" else 1
    gathered = T * P * N * 8 * rows * nvar
    tables = T * P * (4 * rows + (8 * rows if rows == 4 else 0)) + P * 8 * nvar
    data = T * N * 8
    return gathered + tables + data


def cpu_baseline(spec, seconds=12.0):
    """Reference-equivalent CPU path (oracle/beat_oracle.c bo_ffi_seismic_forward, the C
    restatement pinned to the reference) on the host cores of this box, on a bounded
    sample: T_sub of the T targets with the full P x N gather per target."""
    from multiprocessing import get_context

    from oracle import oracle as orc  # noqa: F401  (cpu_baseline leg only)

    T_sub = 4
    ncores = len(os.sched_getaffinity(0))
    rng = np.random.default_rng(spec.seed)
    P, N, S = spec.P, spec.N, spec.S
    ml = spec.interpolation == "multilinear"
    D_cpu = 2 if ml else 1  # D reduced: same gathered volume per step, fits host RAM
    G = rng.standard_normal((T_sub, P, D_cpu, S, N))
    data = rng.standard_normal((T_sub, N))
    w = np.full(T_sub, 2.0)
    slog = np.zeros(T_sub)
    lib_cfg = dict(dur_min=spec.du_min, dur_dt=spec.du_dt, st_min=spec.st_min, st_dt=spec.st_dt)
    fault = dict(ndip=spec.n_patch_dip, nstrike=spec.n_patch_strike, patch_size=spec.patch_size)

    def one(seed):
        r = np.random.default_rng(seed)
        params = dict(slips=r.uniform(0, 5, (1, P)), durations=r.uniform(0.55, 0.7, P) if not ml else r.uniform(0.55, 0.95, P),
                      velocities=r.uniform(*spec.vel_bounds, P),
                      nuc_strike=[r.uniform(6, 13)], nuc_dip=[r.uniform(6, 13)], time=[0.0])
        orc.ffi_seismic_forward([G], lib_cfg, fault, params, data, w, slog, 0.0,
                                interpolation=spec.interpolation, return_synthetics=False)

    global _cpu_one
    _cpu_one = one
    one(0)
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < seconds / 3:
        one(n1)
        n1 += 1
    t1 = (time.perf_counter() - t0) / n1
    rate1 = (T_sub / spec.T) / t1  # full chain-steps/s on one core
    # all cores, one chain per forked worker (iter_parallel_chains, sampler/base.py:428-595)
    rate_all, nsteps = rate1, n1
    if ncores > 1:
        budget = seconds * 2 / 3
        ctx = get_context("fork")
        with ctx.Pool(ncores) as pool:
            pool.map(_cpu_worker, range(ncores))  # warm (page tables of the forked workers)
            t0 = time.perf_counter()
            counts = pool.map(_cpu_worker_for, [(i, budget) for i in range(ncores)])
            dt = time.perf_counter() - t0
        nsteps = int(sum(counts))
        rate_all = nsteps * (T_sub / spec.T) / dt
    return dict(value=rate_all, unit="chain-steps/s", cores=ncores, kind="port",
                value_1core=rate1,
                sample="%d of %d targets (full %dx%d gather per target), %d sample evaluations, "
                       "oracle/beat_oracle.c (C restatement of the reference numpy/C path), "
                       "%d forked workers x 1 thread" % (T_sub, spec.T, P, N, nsteps, ncores))


_cpu_one = None


def _cpu_worker(i):
    _cpu_one(i)
    return 0


def _cpu_worker_for(a):
    """run sample evaluations until the time budget is spent; -> count"""
    i, budget = a
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < budget:
        _cpu_one(1000 * i + k)
        k += 1
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=512, help="chains per GPU per step")
    ap.add_argument("--interp", default="nearest_neighbor",
                    choices=["nearest_neighbor", "multilinear"])
    ap.add_argument("--covariance", default="scalar", choices=["scalar", "toeplitz"])
    ap.add_argument("--targets", type=int, default=64)
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--nstarttimes", type=int, default=25)
    ap.add_argument("--ndurations", type=int, default=3)
    ap.add_argument("--prewhiten", action="store_true",
                    help="toeplitz only: whiten the library once (W.G), no dense W.r per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pmc-summary", default=os.path.join(ROOT, "profiles", "r1_bench_c512_nn_gfstack_summary.json"),
                    help="rocprofv3 PMC summary (tools/run_profile.sh + tools/summarize_rocpd.py) of THIS "
                         "command; supplies roofline.traffic when its configuration matches")
    ap.add_argument("--sort-chains", default=None,
                    help="experiment: order the population by this key before the run (nuc | time)")
    ap.add_argument("--gf-order", type=int, default=None,
                    help="k_gfstack block order: 0 (chain,target,tile) 1 (target,chain,tile)")
    args = ap.parse_args()

    if args.gf_order is not None:
        os.environ["BEATAMD_GF_ORDER"] = str(args.gf_order)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`"
                             % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    # BEATAMD_BENCH_FORCE_DIST=1: exercise the RCCL path with a single rank (1-GPU boxes)
    use_dist = world > 1 or bool(os.environ.get("BEATAMD_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import beat_amd
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population

    dev = torch.device("cuda", local_rank)
    ctx = beat_amd.get_context(local_rank)
    ctx.use_torch_stream()

    spec = SyntheticSpec((20,), (20,), (1.0,), T=args.targets, N=args.samples, D=args.ndurations,
                         S=args.nstarttimes, covariance=args.covariance,
                         interpolation=args.interp, nuc_margin=6.0, time_bounds=(0.0, 0.5))
    t_build = time.perf_counter()
    prob, host = build_problem(spec, device_library=True, ctx=ctx)
    f = prob.compile(ctx, prewhiten="inplace" if args.prewhiten else False)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    B, K, W = args.chains, args.steps, args.warmup
    lay = host["layout"]
    lo, up = lay.bounds(host["lower"], host["upper"])
    # chain c of rank r is global chain r*B + c (seed 1000 + id)
    Q0 = torch.from_numpy(draw_population(spec, lay, host["lower"], host["upper"], B,
                                          seed_offset=1000 + rank * B)).to(dev)
    if args.sort_chains:
        q = Q0.cpu().numpy()
        ns, nd_, tt = (q[:, lay.offset(k)] for k in ("nucleation_strike", "nucleation_dip", "time"))
        if args.sort_chains == "nuc":
            key = np.lexsort((nd_, np.floor(ns / 2.0)))
        elif args.sort_chains in ("pc1", "pc12", "pc1x16"):
            # start-time index field of every chain (what selects the library rows)
            from beat_amd.utility import positions2idxs
            vel = q[:, lay.offset("velocities"):lay.offset("velocities") + spec.P]
            hs = positions2idxs(ns, spec.patch_size[0], min_pos=0.0)
            hd = positions2idxs(nd_, spec.patch_size[0], min_pos=0.0)
            st = ctx.fast_sweep_batch(1.0 / vel, spec.patch_size[0], hs, hd, spec.n_patch_strike[0], spec.n_patch_dip[0])
            sidx = np.rint((st + tt[:, None] - spec.st_min) / spec.st_dt)
            X = sidx - sidx.mean(0)
            U, S_, Vt = np.linalg.svd(X, full_matrices=False)
            pc1, pc2 = U[:, 0] * S_[0], U[:, 1] * S_[1]
            if args.sort_chains == "pc1":
                key = np.argsort(pc1)
            elif args.sort_chains == "pc12":
                r1 = np.argsort(np.argsort(pc1)) // 256      # workgroup by PC1, lanes by PC2
                key = np.lexsort((pc2, r1))
            else:
                r1 = np.argsort(np.argsort(pc1)) // 16       # 16-lane groups by PC1, inside by PC2
                key = np.lexsort((pc2, r1))
        else:
            key = np.argsort(tt)
        Q0 = Q0[torch.from_numpy(key).to(dev)].contiguous()
    L0 = f.batch(Q0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4242 + rank)
    span = torch.from_numpy(up - lo).to(dev)
    # proposal rows: proposal_samples_array[stage_sample] of every chain (metropolis.py:289-313)
    delta = torch.randn((K + W, B, lay.size), generator=gen, device=dev, dtype=torch.float64) \
        * (0.004 * span)
    log_u = torch.log(torch.rand((K + W, B), generator=gen, device=dev, dtype=torch.float64))
    scaling = torch.ones(B, device=dev, dtype=torch.float64)
    lo_d, up_d = torch.from_numpy(lo).to(dev), torch.from_numpy(up).to(dev)
    accepted = torch.zeros(B, device=dev, dtype=torch.int32)
    beta = 2e-6  # an early SMC stage on this problem (tools/smc_app.py): acceptance ~0.2-0.4
    ctx.synchronize()

    def step(i):
        f.astep_batch(Q0, L0, delta[i], scaling, lo_d, up_d, log_u[i], beta, accepted)

    for i in range(W):
        step(i)
    ctx.synchronize()  # also surfaces an out-of-library index as an exception
    ctx.enable_timing(True)
    ctx.reset_timing()
    n_acc = 0
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    ctx.synchronize()
    n_acc = int(accepted.sum().item())

    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    # SMC stage transition exchange (select_end_points, smc.py:188-240): all-gather of the
    # per-rank end points + likelihoods; outside the timed steps, reported separately
    stage_ms = None
    if use_dist:
        from beat_amd import parallel
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        Qall, Lall = parallel.allgather_population(Q0, L0)
        torch.cuda.synchronize()
        stage_ms = (time.perf_counter() - t1) * 1e3
        assert Qall.shape[0] == world * B

    gf_ms, gf_n = ctx.kernel_time("gfstack")
    times = {k: ctx.kernel_time(k) for k in ("sweep", "tables", "grouptables", "gfstack", "quadform",
                                             "finish", "astep")}
    if rank == 0:
        alg = algorithmic_bytes_per_chain_step(spec) * B  # per launch
        rows_per_patch = 4 if spec.interpolation == "multilinear" else 1
        avg_ms = gf_ms / max(gf_n, 1)
        achieved = alg / (avg_ms * 1e-3) / 1e9 if gf_n else 0.0
        out = {
            "metric": "SMC chain-steps/s (FFI seismic gfstacking 400 patches x %d targets x %d samples)"
                      % (spec.T, spec.N),
            "value": world * B * K / dt_max,
            "unit": "chain-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt_max / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: FFI seismic gfstacking, 400 patches x %d targets x "
                            "%d samples, library (%d,%d,%d,%d,%d) f64 = %.1f GB in HBM, %s, "
                            "covariance %s%s" % (spec.T, spec.N, spec.T, spec.P, spec.D, spec.S, spec.N,
                                                 spec.lib_bytes / 1e9, spec.interpolation,
                                                 spec.covariance,
                                                 " (library pre-whitened)" if args.prewhiten else ""),
                "chains_per_gpu": B,
                "global_chains": world * B,
                "parallelism": "chains sharded over %d GPU(s), library replicated" % world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_gfstack" if (B < 48 or os.environ.get("BEATAMD_GF_KERNEL") == "0")
                          else "k_gfstack_shared" if os.environ.get("BEATAMD_GS_DMA") == "0"
                          else "k_gfstack_dma",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_launch": alg,
                "avg_launch_ms": avg_ms,
                "launches": gf_n,
                # frac > 1 on the chain-shared kernels: `achieved` credits no cross-chain reuse
                #  while rows shared by chains are fetched once (`traffic`).  What
                # binds those kernels is the per-lane LDS gather: one 8-byte LDS operand per FMA.
                "lds_gather": {
                    "bytes_per_launch": float(B) * spec.T * spec.P * spec.N * 8 * rows_per_patch,
                    "floor_ms": float(B) * spec.T * spec.P * spec.N * 8 * rows_per_patch
                                / (256.0 * 256 * 2.4e9) * 1e3,
                    "ceiling_ms_measured": 3.53 * (float(B) * spec.T * spec.P * spec.N * rows_per_patch)
                                           / (512.0 * 64 * 400 * 4096),
                    "source": "tools/micro/ldsgather.hip (profiles/r1_variants.md)",
                },
            },
            "kernel_ms_per_step": {k: (v[0] / K) for k, v in times.items() if v[1]},
            "accept_rate_last_step": n_acc / float(B),
            "setup_s": t_build,
        }
        # HBM traffic of the dominant kernel from the PMC passes of the same command
        # (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streams, +WRITE_SIZE)
        default_cfg = (B == 512 and spec.interpolation == "nearest_neighbor" and spec.covariance == "scalar"
                       and spec.T == 64 and spec.N == 4096 and args.gf_order is None
                       and not any(k.startswith("BEATAMD_G") for k in os.environ))
        if default_cfg and os.path.exists(args.pmc_summary):
            pmc = json.load(open(args.pmc_summary))
            if "hbm_read_bytes_per_launch_corrected" in pmc:
                out["roofline"]["traffic"] = (pmc["hbm_read_bytes_per_launch_corrected"]
                                              + pmc.get("hbm_write_bytes_per_launch", 0.0))
                out["roofline"]["traffic_source"] = os.path.relpath(args.pmc_summary, ROOT)
        if stage_ms is not None:
            out["stage_transition_ms"] = stage_ms
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
