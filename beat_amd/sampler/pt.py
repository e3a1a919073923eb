"""
Parallel tempering -- counterpart of beat/sampler/pt.py.

The reference runs one MPI rank per tempered chain in a master/worker star: workers
sample a random number of Metropolis steps, send their last point to the master, which
pairs the first two arrivals and applies ``propose_chain_swap`` (:429-457).  Here all
replicas of a rank advance in ONE batched device call per step (per-replica beta), and the
exchange is a synchronous round: an all-gather of one likelihood per replica (RCCL), the same
even/odd adjacent-temperature swap sweep on every rank with a shared RandomState, and the swapped
states moved where they live (device gather; point to point only across rank boundaries).  The
swap rule, the beta ladder (``update_betas`` :179-221), the beta tuning (``tune_betas``
:331-354, ``tune`` :37-73) and the swap interval draw are the reference's; the pairing
order differs (asynchronous first-arrivals are not reproducible) -- statistical, not bit,
parity.
"""
import numpy as np

from .. import parallel
from .metropolis import BatchedMetropolis


def tune(scale, acc_rate):
    """pt.py:37-73"""
    if acc_rate < 0.001:
        scale *= 0.85
    elif acc_rate < 0.05:
        scale *= 0.9
    elif acc_rate < 0.2:
        scale *= 0.95
    elif acc_rate > 0.95:
        scale *= 1.15
    elif acc_rate > 0.75:
        scale *= 1.10
    elif acc_rate > 0.5:
        scale *= 1.05
    return scale


def propose_chain_swap(beta1, beta2, llk1, llk2, log_u):
    """pt.py:429-457: alpha = (beta2 - beta1) * (llk1 - llk2); accept iff log(u) < alpha"""
    return log_u < (beta2 - beta1) * (llk1 - llk2)


class TemperingManager(object):
    """Ladder and swap bookkeeping (pt.py:101-470) for ``n_ladders`` temperatures x
    ``n_replicas`` independent replicas per temperature (BASELINE config 5: 32 x 256).
    Replica r of temperature k is global chain k * n_replicas + r."""

    def __init__(self, n_workers_posterior, n_workers_tempered, n_replicas=1,
                 swap_interval=(100, 300), beta_tune_interval=1000, random_seed=17):
        self.n_workers_posterior = int(n_workers_posterior)
        self.n_workers_tempered = int(n_workers_tempered)
        self.n_workers = self.n_workers_posterior + self.n_workers_tempered
        self.n_replicas = int(n_replicas)
        self.swap_interval = swap_interval
        self.beta_tune_interval = beta_tune_interval
        self._t_scale_min, self._t_scale_max = 1.01, 2.0
        self.current_scale = 1.2
        self.rng = np.random.RandomState(random_seed)  # shared by all ranks
        self.acceptance_matrix = np.zeros((self.n_workers, self.n_workers), dtype="int64")
        self.sample_count = np.zeros_like(self.acceptance_matrix)
        self.history = []
        self.update_betas()
        self._round = 0

    def update_betas(self, t_scale=None):
        """pt.py:179-221: betas = [1]*n_posterior + t_scale**-k, k = 1..n_tempered"""
        if t_scale is None:
            t_scale = self.current_scale
        self.current_scale = t_scale
        temperature = np.power(t_scale, np.arange(1, self.n_workers_tempered + 1))
        self.betas = np.array([1.0] * self.n_workers_posterior + (1.0 / temperature).tolist())
        return self.betas

    @property
    def chain_betas(self):
        return np.repeat(self.betas, self.n_replicas)

    def draw_swap_interval(self):
        """DiscreteBoundedUniform(lower, upper) (pt.py:149-152, config.py:1728-1738)"""
        lo, up = self.swap_interval
        return int(self.rng.randint(low=up - lo) + lo) if up > lo else int(lo)

    def swap_round(self, like):
        """One synchronous exchange sweep.  like (n_workers*n_replicas,) -> permutation p with
        new_state[i] = old_state[p[i]].  Even rounds pair ladders (0,1),(2,3).., odd rounds
        (1,2),(3,4)..; every replica column is an independent ladder."""
        K, R = self.n_workers, self.n_replicas
        perm = np.arange(K * R)
        like = np.asarray(like).reshape(K, R)
        start = self._round % 2
        logu = np.log(self.rng.uniform(size=(K, R)))
        for k in range(start, K - 1, 2):
            acc = propose_chain_swap(self.betas[k], self.betas[k + 1], like[k], like[k + 1], logu[k])
            self.sample_count[k, k + 1] += R
            self.sample_count[k + 1, k] += R
            n = int(acc.sum())
            self.acceptance_matrix[k, k + 1] += n
            self.acceptance_matrix[k + 1, k] += n
            a = k * R + np.nonzero(acc)[0]
            b = a + R
            perm[a], perm[b] = b, a
        self._round += 1
        return perm

    def get_acceptance_swap(self, k):
        c = self.sample_count[k, k + 1]
        return self.acceptance_matrix[k, k + 1] / float(c) if c else 0.0

    def tune_betas(self):
        """pt.py:331-354: acceptance between the posterior level and the first tempered
        level drives the temperature scale (inverse behaviour of step scaling)."""
        k = max(self.n_workers_posterior - 1, 0)
        acceptance = self.get_acceptance_swap(k)
        t_scale = min(max(tune(self.current_scale, acceptance), self._t_scale_min), self._t_scale_max)
        self.history.append((self.current_scale, acceptance))
        self.acceptance_matrix[:] = 0
        self.sample_count[:] = 0
        return self.update_betas(t_scale)


def pt_sample(target, lower, upper, n_chains_posterior=1, n_chains_tempered=7, n_replicas=1,
              n_samples=1000, swap_interval=(100, 300), beta_tune_interval=10, proposal_cov=None,
              device=None, random_seed=17, tune_interval=100, record_every=1):
    """pt.py:793-906 driver.  Returns (posterior samples (n, nparams), their likelihood
    vectors, manager).  Samples of the beta == 1 replicas are recorded after every round.

    Exchange round: all-gather of ONE likelihood per replica, the swap sweep decided
    identically on every rank (shared RandomState), then the permutation is applied where the
    states live: local rows by a device gather, rows that change rank point to point
    (``parallel.exchange_rows``).  The replica states themselves are never gathered."""
    import torch
    rank, world, _ = parallel.ensure_group()
    man = TemperingManager(n_chains_posterior, n_chains_tempered, n_replicas, swap_interval,
                           beta_tune_interval, random_seed)
    n_total = man.n_workers * n_replicas
    start, stop = parallel.chain_block(n_total, rank, world)
    lower, upper = np.asarray(lower, dtype=np.float64), np.asarray(upper, dtype=np.float64)
    stepper = BatchedMetropolis(target, lower, upper, stop - start, device=device,
                                tune_interval=tune_interval, seed=random_seed, first_chain=start)
    dev = stepper.device
    ops = stepper.ops
    if proposal_cov is None:
        proposal_cov = np.diag(((upper - lower) * 0.05) ** 2)
    stepper.set_proposal(proposal_cov)
    Qall = lower + (upper - lower) * man.rng.random_sample((n_total, lower.size))
    Q = torch.from_numpy(np.ascontiguousarray(Qall[start:stop])).to(dev)
    L = stepper.evaluate(Q).to(dev)
    npar = Q.shape[1]
    # posterior replicas (beta == 1) are the first n_posterior * n_replicas global chains
    n_post = man.n_workers_posterior * n_replicas
    p0, p1 = min(start, n_post), min(stop, n_post)
    samples, lsamples = [], []
    rounds = 0
    # wall time and step count of the rounds alone (bench.py's pt_leg): man.loop_seconds / man.loop_steps
    if Q.is_cuda:
        torch.cuda.synchronize(dev)
    import time
    t_loop, man.loop_steps = time.perf_counter(), 0
    while sum(len(s) for s in samples) < n_samples:
        betas = torch.from_numpy(man.chain_betas[start:stop].copy()).to(dev)
        for _ in range(man.draw_swap_interval()):
            stepper.step(Q, L, betas)
            man.loop_steps += 1
        ops.check()
        QL = torch.cat([Q, L], 1)
        if rounds % record_every == 0:
            post = parallel.allgather_rows(QL[p0 - start:p1 - start]).cpu().numpy()
            samples.append(post[:, :npar].copy())
            lsamples.append(post[:, npar:].copy())
        like = parallel.allgather_rows(L[:, -1:].contiguous(), n_total)[:, 0].cpu().numpy()
        perm = man.swap_round(like)
        QL = parallel.exchange_rows(QL, perm, n_total, gather=ops.gather)
        Q, L = QL[:, :npar].contiguous(), QL[:, npar:].contiguous()
        rounds += 1
        if rounds % beta_tune_interval == 0 and man.n_workers_tempered > 0:
            man.tune_betas()
    if Q.is_cuda:
        torch.cuda.synchronize(dev)
    man.loop_seconds = time.perf_counter() - t_loop
    return np.concatenate(samples)[:n_samples], np.concatenate(lsamples)[:n_samples], man
