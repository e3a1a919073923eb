"""
Sequential Monte Carlo (CATMIP / TMCMC) -- counterpart of beat/sampler/smc.py.

Same stage logic and method names as the reference (``calc_beta`` :133-165,
``calc_covariance`` :167-186, ``select_end_points`` :188-240, ``resample`` :290-324,
``smc_sample`` :333-546).  What changes is where the work runs:

  * all chains of a rank advance together on the GPU (``BatchedMetropolis``): the
    reference's fork pool over chains (sampler/base.py:428-595) becomes the batch
    dimension of the kernels;
  * chains are sharded over ranks in contiguous blocks; at every stage boundary the end
    points and likelihood vectors are all-gathered (RCCL over xGMI) and every rank then
    computes beta, the weighted covariance and the resampling IDENTICALLY from the same
    arrays and the same seeded draw -- no rank-0 bottleneck, no files.
"""
import logging
import os

import numpy as np

from .. import parallel
from ..utility import ensure_cov_psd
from .metropolis import BatchedMetropolis

logger = logging.getLogger("smc")

sample_factor_final_stage = 1  # smc.py:24


class SMC(object):
    """Stage state of the sampler; attribute names follow the reference class."""

    def __init__(self, target, lower, upper, n_chains=100, tune=True, tune_interval=100,
                 coef_variation=1.0, check_bound=True, proposal_name="MultivariateNormal",
                 device=None, random_seed=42, scale=1.0):
        if proposal_name not in ("MultivariateNormal", "MultivariateCauchy"):
            raise NotImplementedError("the GPU SMC uses the MultivariateNormal / MultivariateCauchy proposals")
        if not check_bound:
            raise NotImplementedError("check_bound=False is not supported")
        self.target = target
        self.lower = np.asarray(lower, dtype=np.float64)
        self.upper = np.asarray(upper, dtype=np.float64)
        self.n_chains = int(n_chains)
        self.coef_variation = coef_variation
        self.proposal_name = proposal_name
        self.beta, self.old_beta = 0.0, 0.0
        self.stage = 0
        self.likelihoods = np.zeros(self.n_chains)
        self.weights = None
        self.covariance = None
        self.resampling_indexes = np.arange(self.n_chains)
        self.array_population = None       # (n_chains, nparams) end points, all ranks
        self.array_lpoints = None          # (n_chains, nllk) likelihood vectors, all ranks
        self.rng = np.random.RandomState(random_seed)  # identical on every rank
        self.rank, self.world, _ = parallel.dist_info()
        self.block = parallel.chain_block(self.n_chains, self.rank, self.world)
        n_local = self.block[1] - self.block[0]
        self.stepper = BatchedMetropolis(target, lower, upper, n_local, device=device, tune=tune,
                                         tune_interval=tune_interval, scale=scale,
                                         seed=random_seed + 1000003 * self.rank)
        self.device = self.stepper.device
        self.n_steps = 1
        self.stage_betas, self.stage_acceptance = [], []

    # ------------------------------------------------------------------ population
    def initialize_population(self):
        """metropolis.py:125-152: prior draws (Uniform boxes) for every chain; identical on
        every rank because the seeded RandomState is shared."""
        u = self.rng.random_sample((self.n_chains, self.lower.size))
        self.array_population = self.lower + (self.upper - self.lower) * u
        return self.array_population

    # ------------------------------------------------------------------ reference methods
    def calc_beta(self):
        """smc.py:133-165"""
        low_beta, up_beta, old_beta = self.beta, 2.0, self.beta
        current_beta = self.beta
        while up_beta - low_beta > 1e-6:
            current_beta = (low_beta + up_beta) / 2.0
            temp = np.exp((current_beta - self.beta) * (self.likelihoods - self.likelihoods.max()))
            cov_temp = np.std(temp) / np.mean(temp)
            if cov_temp > self.coef_variation:
                up_beta = current_beta
            else:
                low_beta = current_beta
        weights = temp / np.sum(temp)
        return current_beta, old_beta, weights

    def calc_covariance(self, repair=True):
        """smc.py:167-186.  repair=False skips ensure_cov_psd on the host: the device proposal
        repairs and factors the matrix in one eigendecomposition."""
        cov = np.cov(self.array_population, aweights=self.weights.ravel(), bias=False, rowvar=0)
        cov = np.atleast_2d(cov)
        if repair:
            cov = ensure_cov_psd(cov)
        if np.isnan(cov).any() or np.isinf(cov).any():
            raise ValueError("Sample covariances contains Inf or NaN! Please try reducing the"
                             " upper and lower bounds of hyper parameters!")
        return cov

    def set_stage_proposal(self):
        """Proposal of the coming stage: N(0, weighted population covariance) (smc.py:452-466).
        Populations up to twice the parameter count draw straight from the weighted, centred
        population (no factorisation of a singular / ill-conditioned matrix); larger ones
        factor ``self.covariance`` once."""
        if self.n_chains <= 2 * self.array_population.shape[1]:
            self.stepper.set_proposal_from_population(self.array_population, self.weights, self.proposal_name)
        else:
            self.stepper.set_proposal(self.covariance, self.proposal_name)

    def resample(self):
        """smc.py:290-324 Kitagawa's deterministic resampling; the single auxiliary draw
        comes from the rank-shared RandomState.  Guarded against cumulative-sum overrun
        (SURVEY A.15)."""
        parents = np.arange(self.n_chains)
        N_childs = np.zeros(self.n_chains, dtype=int)
        cum_dist = np.cumsum(self.weights)
        aux = self.rng.rand(1)
        u = (parents + aux) / self.n_chains
        j = 0
        for i in parents:
            while u[i] > cum_dist[j] and j < self.n_chains - 1:
                j += 1
            N_childs[j] += 1
        return np.repeat(parents, N_childs)

    def select_end_points(self, Q_local, L_local):
        """smc.py:188-240 -- instead of reading trace files: all-gather the ranks' blocks."""
        Qall, Lall = parallel.allgather_population(Q_local, L_local)
        self.array_population = Qall.detach().cpu().numpy()
        self.array_lpoints = Lall.detach().cpu().numpy()
        self.likelihoods = self.array_lpoints[:, -1].copy()
        return self.array_population, self.likelihoods

    def get_map_end_points(self):
        """smc.py:277-288"""
        return self.array_population[self.likelihoods.flatten().argmax(), :]

    # ------------------------------------------------------------------ sampling of one stage
    def _local(self, arr):
        import torch
        a = np.ascontiguousarray(arr[self.block[0]:self.block[1]])
        return torch.from_numpy(a).to(self.device)

    def sample_stage(self, n_steps, on_step=None):
        """iter_parallel_chains for one stage: every local chain starts at its resampled
        parent (sampler/base.py:541-571) and takes n_steps Metropolis steps at self.beta."""
        start = self.array_population[self.resampling_indexes]
        lstart = self.array_lpoints[self.resampling_indexes]
        Q, L = self._local(start), self._local(lstart)
        import torch
        n_acc = torch.zeros((), dtype=torch.int64, device=Q.device)  # stays on the device
        for i in range(int(n_steps)):
            acc = self.stepper.step(Q, L, self.beta)
            n_acc += acc.sum()
            if on_step is not None:
                on_step(i, Q, L, acc)
        self.stage_acceptance.append(float(n_acc.item()) / max(1.0, float(n_steps) * Q.shape[0]))
        return Q, L


def _dump_stage(step, homepath, layout, out_names, backend):
    """stage directory with one-draw traces of every chain (rank 0 writes; every rank holds the
    same gathered arrays) + the sampler state needed to resume (smc.py:549-557)"""
    if homepath is None or step.rank != 0:
        return
    from ..backend import stage_path, write_population
    path = write_population(homepath, step.stage, layout, out_names, step.array_population,
                            step.array_lpoints, backend)
    np.savez(os.path.join(path, "sampler_state.npz"), beta=step.beta, old_beta=step.old_beta,
             stage=step.stage, population=step.array_population, lpoints=step.array_lpoints,
             covariance=step.covariance if step.covariance is not None else np.zeros(0),
             scaling=step.stepper.scaling.cpu().numpy())


def load_stage(step, homepath, stage):
    """resume: restore the population and tempering state of a completed stage
    (init_stage / load_sampler_params, sampler/base.py:618-661, backend.py:1049-1071)"""
    from ..backend import stage_path
    z = np.load(os.path.join(stage_path(homepath, stage), "sampler_state.npz"))
    step.beta, step.old_beta, step.stage = float(z["beta"]), float(z["old_beta"]), int(z["stage"])
    step.array_population, step.array_lpoints = z["population"], z["lpoints"]
    step.likelihoods = step.array_lpoints[:, -1].copy()
    return step


def smc_sample(n_steps, step, progressbar=False, on_stage=None, max_stages=200, homepath=None,
               layout=None, out_names=None, backend="bin", resume_stage=None):
    """smc.py:333-546 stage loop.  Returns the final population (n_chains, nparams), the
    likelihood vectors and the list of betas.  With ``homepath`` every stage leaves a
    ``stage_<k>`` / ``stage_final`` directory of NumpyChain/TextChain traces (beat_amd.backend)."""
    import torch
    step.n_steps = int(n_steps)
    if n_steps < 1:
        raise TypeError("Argument `n_steps` should be above 0.")
    if resume_stage is not None:
        load_stage(step, homepath, resume_stage)
    else:
        # stage 0: evaluate the prior population (draws = 1, no move)
        if step.array_population is None:
            step.initialize_population()
        Q = step._local(step.array_population)
        L = step.stepper.evaluate(Q)
        step.select_end_points(Q, L if torch.is_tensor(L) else torch.from_numpy(np.asarray(L)))
        if layout is not None:
            _dump_stage(step, homepath, layout, out_names, backend)
    betas = [step.beta]
    while step.beta < 1.0 and step.stage < max_stages:
        step.beta, step.old_beta, step.weights = step.calc_beta()
        if step.beta > 1.0:
            step.beta = 1.0
            break
        step.covariance = step.calc_covariance(repair=False)
        step.set_stage_proposal()
        step.resampling_indexes = step.resample()
        step.stage += 1
        logger.info("Beta: %f Stage: %i", step.beta, step.stage)
        Q, L = step.sample_stage(n_steps)
        step.select_end_points(Q, L)
        betas.append(step.beta)
        if layout is not None:
            _dump_stage(step, homepath, layout, out_names, backend)
        if on_stage is not None:
            on_stage(step)
    # final stage at beta = 1 (smc.py:526-543)
    step.stage = -1
    temp = np.exp((1 - step.old_beta) * (step.likelihoods - step.likelihoods.max()))
    step.weights = temp / np.sum(temp)
    step.covariance = step.calc_covariance(repair=False)
    step.set_stage_proposal()
    step.resampling_indexes = step.resample()
    step.beta = 1.0
    Q, L = step.sample_stage(n_steps * sample_factor_final_stage)
    step.select_end_points(Q, L)
    betas.append(1.0)
    if layout is not None:
        _dump_stage(step, homepath, layout, out_names, backend)
    step.stage_betas = betas
    return step.array_population, step.array_lpoints, betas


def tune(acc_rate):
    """smc.py:558-575 (Muto & Beck 2008)"""
    a, b = 1.0 / 9, 8.0 / 9
    return np.power((a + (b * acc_rate)), 2)
