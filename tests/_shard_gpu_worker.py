"""Target-sharded libraries on 1 or 2 ranks (tests/test_gpu_dist.py): BEATAMD_TEST_MODE = "replicated" (one rank, the
whole model) or "targets" (every rank compiles the rows of ITS targets only, beat_amd.models.sharded).  Two ranks share
the one GPU of the box through gloo (collectives staged through host memory) unless BEATAMD_TEST_BACKEND=nccl.
Evaluates a population, runs a few SMC stages with all chains on every rank and writes rank 0's results; the test
compares replicated and sharded runs (reference semantics: beat/models/seismic.py:1332-1349 one logpt per dataset,
beat/models/problems.py:227-247 like = sum of the composites)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    import beat_amd
    from beat_amd import parallel
    from beat_amd.models.sharded import TargetShardedLogp
    from beat_amd.sampler import SMC, smc_sample
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population

    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("BEATAMD_TEST_BACKEND", "gloo")
    mode = os.environ.get("BEATAMD_TEST_MODE", "replicated")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    rank = 0
    if world > 1:
        os.environ["LOCAL_RANK"] = str(local)
        rank, world, _ = parallel.init(backend)
    dev = torch.device("cuda", local)
    ctx = beat_amd.get_context(local)
    # 5 targets over 2 ranks (3 + 2), two slip variables, station shifts, dense covariance, geodetic + Laplacian replicated
    spec = SyntheticSpec((5,), (5,), (1.0,), T=5, N=96, D=3, S=25, covariance="toeplitz", slip_varnames=("uparr", "uperp"),
                         station_shifts=True, geodetic_nobs=(20, 31), laplacian=True, interpolation="multilinear")
    prob, host = build_problem(spec)
    lay = host["layout"]
    lo, up = lay.bounds(host["lower"], host["upper"])
    f = TargetShardedLogp(prob, ctx) if mode == "targets" else prob.compile(ctx)
    if mode == "targets":
        assert f.world == world and f.nllk == 5 + 2 + 1 + 1 and sum(f.n_local) == {1: [5], 2: [3, 2]}[world][rank]
    Q = torch.from_numpy(draw_population(spec, lay, host["lower"], host["upper"], 300)).to(dev)
    Q[7, lay.offset("durations")] = 99.0           # a chain outside the library grid: like = NaN on every rank
    LL = f.batch(Q)
    try:
        ctx.synchronize()      # the out-of-library index is reported once (IndexError, like the reference) ...
        raise AssertionError("expected the IndexError of the chain outside the library grid")
    except IndexError:
        pass                   # ... and leaves NaN in that chain's `like`
    if mode == "targets":
        # a start time that leaves the grid on the targets of ONE rank only -- a station correction of
        # a station whose channels all live on the last rank (targets 3, 4) -- sets the status word there alone; the
        # collective check makes EVERY rank raise the IndexError instead of leaving the others in the next all-gather
        prob2, host2 = build_problem(spec)
        name, _ = host2["time_shifts"]
        prob2.wavemaps[0].time_shifts = (name, np.array([0, 0, 0, 1, 1]))
        f2 = TargetShardedLogp(prob2, ctx)
        Q2 = torch.from_numpy(draw_population(spec, lay, host["lower"], host["upper"], 64)).to(dev)
        Q2[11, lay.offset(name, 1)] = -500.0
        LL2 = f2.batch(Q2)
        try:
            f2.check_collectively()
            raise AssertionError("rank %d: expected the IndexError raised on the rank that owns targets 3, 4" % rank)
        except IndexError:
            pass
        assert bool(torch.isnan(LL2[11, -1])) and int(torch.isnan(LL2[:, -1]).sum()) == 1
        f2.check_collectively()           # (the status words were cleared: a healthy evaluation passes on every rank)
        # the step in pieces (propose / forward + gather + assemble / accept kernels) = the same step of the local torch twin
        Qs = Q2[32:64].contiguous()        # (chain 11 carries NaN: kept out of the bitwise comparisons below)
        L0 = f2.batch(Qs).clone()
        q0, l0 = Qs.clone(), L0.clone()
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        delta = torch.randn(q0.shape, generator=g, device=dev, dtype=torch.float64) * 1e-3
        log_u = torch.log(torch.rand(32, generator=g, device=dev, dtype=torch.float64))
        sc = torch.ones(32, device=dev, dtype=torch.float64)
        lo_t, up_t = torch.from_numpy(lo).to(dev), torch.from_numpy(up).to(dev)
        acc = f2.astep_batch(q0, l0, delta, sc, lo_t, up_t, log_u, 0.3)
        qp = Qs + delta
        inb = ((qp >= lo_t) & (qp <= up_t)).all(1)
        lp = f2.batch(torch.where(inb[:, None], qp, Qs).contiguous())
        mr = 0.3 * (lp[:, -1] - L0[:, -1])
        want = inb & torch.isfinite(mr) & (log_u < mr)
        assert torch.equal(acc.bool(), want), (int(want.sum()), int(acc.sum()))
        assert torch.equal(q0, torch.where(want[:, None], qp, Qs)) and torch.equal(l0, torch.where(want[:, None], lp, L0))
        f2.release()
    step = SMC(f, lo, up, n_chains=256, device=dev, random_seed=11, tune_interval=3,
               shard="targets" if mode == "targets" else "chains")
    pop, lp, betas = smc_sample(4, step, max_stages=3, final_stage=False)
    if mode == "targets" and world > 1:
        both = parallel.allgather_rows(torch.stack([step.Q_all.sum(), step.L_all.sum()])[None])
        assert torch.equal(both[0], both[1]), both
    if rank == 0:
        np.savez(os.environ["BEATAMD_TEST_OUT"], LL=LL.cpu().numpy(), pop=pop, lp=lp, betas=np.asarray(betas))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    print("SHARD_GPU_WORKER_OK rank", rank, flush=True)


if __name__ == "__main__":
    main()
