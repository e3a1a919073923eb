"""Device samplers on 1 or more ranks (tests/test_gpu_dist.py).

BEATAMD_TEST_BACKEND=gloo (default): every rank uses cuda:0 -- two ranks share ONE GPU, the
collectives go through host memory (beat_amd.parallel._staged); RCCL refuses duplicate devices.
BEATAMD_TEST_BACKEND=nccl: one GPU per rank, RCCL (boxes with >= 2 GPUs).

Runs SMC (a few stages) and parallel tempering (a few exchange rounds) of a small FFI problem with
DeviceOps and writes rank 0's results to BEATAMD_TEST_OUT; the test compares the files of a
1-rank and a 2-rank run bit for bit (reference semantics: beat/sampler/smc.py:188-240,
beat/sampler/pt.py:429-457, 573-633)."""
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
    from beat_amd.sampler import SMC, pt_sample, smc_sample
    from beat_amd.sampler.ops import DeviceOps
    from beat_amd.synthetic import SyntheticSpec, build_problem

    os.environ["BEATAMD_CHECK_RANKS"] = "1"   # transition(): beta, weights, indices, factor bitwise equal on all ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("BEATAMD_TEST_BACKEND", "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    rank = 0
    if world > 1:
        os.environ["LOCAL_RANK"] = str(local)
        rank, world, _ = parallel.init(backend)
    dev = torch.device("cuda", local)
    ctx = beat_amd.get_context(local)
    spec = SyntheticSpec((5,), (5,), (1.0,), T=3, N=64, D=3, S=25, covariance="toeplitz")
    prob, host = build_problem(spec)
    f = prob.compile(ctx)
    lo, up = host["layout"].bounds(host["lower"], host["upper"])

    # ---- SMC, 256 chains over the ranks: end points all-gathered per stage, stage decisions on every rank
    step = SMC(f, lo, up, n_chains=256, device=dev, random_seed=11, tune_interval=3)
    assert isinstance(step.ops, DeviceOps) and step.block == parallel.chain_block(256, rank, world)
    checks = []

    def on_stage(s):
        # every rank holds the same gathered population, weights, restart indices and proposal factor
        chk = torch.stack([s.Q_all.sum(), s.L_all.sum(), s.w.sum(), s.idx.double().sum(),
                           s.stepper.factor.sum(), torch.tensor(s.beta, device=dev, dtype=torch.float64)])
        checks.append(chk)
    pop, lp, betas = smc_sample(4, step, max_stages=3, on_stage=on_stage)
    if world > 1:
        for chk in checks:
            both = parallel.allgather_rows(chk[None])
            assert torch.equal(both[0], both[1]), (both[0], both[1])

    # ---- parallel tempering: 4 temperatures x 32 replicas over the ranks; adjacent temperatures meet
    # across the rank boundary (rows move point to point)
    s, ls, man = pt_sample(f, lo, up, n_chains_posterior=1, n_chains_tempered=3, n_replicas=32,
                           n_samples=32 * 4, swap_interval=(3, 5), beta_tune_interval=2, device=dev,
                           random_seed=5, tune_interval=4)
    if rank == 0:
        np.savez(os.environ["BEATAMD_TEST_OUT"], pop=pop, lp=lp, betas=np.asarray(betas), s=s, ls=ls,
                 scale=man.current_scale, nstage_checks=len(checks))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    print("DIST_GPU_WORKER_OK rank", rank, flush=True)


if __name__ == "__main__":
    main()
