"""world_size-2 worker (gloo, CPU) for tests/test_distributed_cpu.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    os.environ["BEATAMD_CHECK_RANKS"] = "1"   # SMC.transition asserts bitwise equal decisions on all ranks
    import torch
    import torch.distributed as dist

    from beat_amd import parallel
    from beat_amd.sampler import SMC, pt_sample, smc_sample
    from beat_amd.sampler.hosttarget import HostTarget
    from test_samplers_cpu import _two_gaussians

    rank, world, _ = parallel.init("gloo")
    assert world == 2

    # uneven blocks: 7 chains over 2 ranks -> 4 + 3, global order preserved
    a, b = parallel.chain_block(7, rank, world)
    Q = torch.arange(a, b, dtype=torch.float64)[:, None] * torch.ones(1, 3, dtype=torch.float64)
    L = -torch.arange(a, b, dtype=torch.float64)[:, None]
    Qall, Lall = parallel.allgather_population(Q, L)
    assert Qall.shape == (7, 3) and torch.equal(Qall[:, 0], torch.arange(7, dtype=torch.float64))
    assert torch.equal(Lall[:, 0], -torch.arange(7, dtype=torch.float64))

    # replica exchange: moving only the rows that change place equals the all-gather + permute of
    # the round-1 implementation, for arbitrary permutations and uneven blocks
    rs = np.random.RandomState(77)
    for n_total in (7, 32, 33):
        a, b = parallel.chain_block(n_total, rank, world)
        Xg = torch.from_numpy(rs.standard_normal((n_total, 5)))
        for trial in range(4):
            perm = rs.permutation(n_total) if trial < 2 else np.arange(n_total)
            if trial == 3:   # adjacent swaps across the block boundary only
                b0 = parallel.chain_block(n_total, 0, world)[1]
                perm[[b0 - 1, b0]] = perm[[b0, b0 - 1]]
            got = parallel.exchange_rows(Xg[a:b].clone(), perm, n_total)
            ref = parallel.allgather_rows(Xg[a:b].clone())[torch.from_numpy(perm)][a:b]
            assert torch.equal(got, ref), (n_total, trial)

    # SMC with chains sharded over the 2 ranks
    f, n = _two_gaussians()
    step = SMC(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains=301, tune_interval=10,
               random_seed=3)
    assert step.block == parallel.chain_block(301, rank, world)
    pop, lp, betas = smc_sample(60, step)
    # every rank holds the identical gathered population and made identical stage decisions
    chk = torch.tensor([pop.sum(), lp.sum(), float(len(betas)), betas[1]], dtype=torch.float64)
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    assert torch.equal(both[0], both[1]), (both[0], both[1])
    assert np.allclose(np.abs(pop).mean(axis=0), 0.5, atol=0.04), np.abs(pop).mean(axis=0)

    # PT: 8 ladders x 4 replicas over 2 ranks
    s, ls, man = pt_sample(HostTarget(f, n), -2 * np.ones(n), 2 * np.ones(n), n_chains_posterior=2,
                           n_chains_tempered=6, n_replicas=4, n_samples=600, swap_interval=(20, 40),
                           beta_tune_interval=5, proposal_cov=np.eye(n) * 0.02, random_seed=5)
    chk = torch.tensor([s.sum(), man.current_scale], dtype=torch.float64)
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    assert torch.equal(both[0], both[1])
    assert np.allclose(np.abs(s[200:]).mean(axis=0), 0.5, atol=0.08)

    x = parallel.broadcast_array(np.array([rank + 5.0]), src=1)
    assert x[0] == 6.0
    dist.barrier()
    dist.destroy_process_group()
    print("DIST_WORKER_OK rank", rank)


if __name__ == "__main__":
    main()
