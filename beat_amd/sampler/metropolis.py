"""
Metropolis stepping -- counterpart of beat/sampler/metropolis.py.

``Metropolis.astep`` keeps the reference's one-chain semantics (:276-422) on top of any
compiled log-likelihood function; ``BatchedMetropolis`` advances all chains of a rank
with one fused device call per step (``beatamd_ffi_astep_batch``), which is what the
SMC / PT drivers use.
"""
import numpy as np

from .base import (DeviceMvNormalProposal, choose_proposal, metrop_select,  # noqa: F401
                   multivariate_proposals, step_tune)


class Metropolis(object):
    """One-chain Metropolis step with the reference's attributes and decision sequence.

    logp_forw_func   callable q -> list of arrays, last entry = ``like``  (sampler/base.py:598-615)
    prior_logp_func  callable q -> 0-d array, finite inside the prior box (metropolis.py:176-181)
    """

    def __init__(self, logp_forw_func, prior_logp_func, nparams, scale=1.0, n_chains=100, tune=True,
                 tune_interval=100, check_bound=True, proposal_name="MultivariateNormal",
                 proposal_scale=None):
        self.logp_forw_func = logp_forw_func
        self.prior_logp_func = prior_logp_func
        self.scaling = np.float64(scale)
        self.tune = tune
        self.check_bound = check_bound
        self.tune_interval = tune_interval
        self.steps_until_tune = tune_interval
        self.stage_sample = 0
        self.cumulative_samples = 0
        self.accepted = 0
        self.beta = 1.0
        self.stage = 0
        self.chain_index = 0
        self.n_chains = n_chains
        self.n_steps = 1
        self._llk_index = -1
        self.proposal_name = proposal_name
        if proposal_scale is None:
            proposal_scale = np.eye(nparams) if proposal_name in multivariate_proposals \
                else np.ones(nparams)
        self.proposal_dist = choose_proposal(proposal_name, scale=proposal_scale)
        self.chain_previous_lpoint = [[]] * n_chains

    def astep(self, q0):
        """metropolis.py:276-422 (continuous variables)"""
        if self.stage == 0:
            l_new = self.logp_forw_func(q0)
            if not np.isfinite(l_new[self._llk_index]):
                raise ValueError("Got NaN in likelihood evaluation! Invalid model definition? "
                                 "Or starting point outside prior bounds!")
            return q0, l_new
        if self.stage_sample == 0:
            self.proposal_samples_array = self.proposal_dist(self.n_steps).astype("float64")
        if not self.steps_until_tune and self.tune:
            self.scaling = np.float64(step_tune(self.scaling, self.accepted / float(self.tune_interval)))
            self.steps_until_tune = self.tune_interval
            self.accepted = 0
        delta = self.proposal_samples_array[self.stage_sample, :] * self.scaling
        q = q0 + delta
        l0 = self.chain_previous_lpoint[self.chain_index]
        if len(l0) == 0:
            l0 = self.logp_forw_func(q0)
            self.chain_previous_lpoint[self.chain_index] = l0
        q_new, l_new = q0, l0
        if (not self.check_bound) or np.isfinite(self.prior_logp_func(q)):
            lp = self.logp_forw_func(q)
            q_new, accepted = metrop_select(
                self.beta * (lp[self._llk_index] - l0[self._llk_index]), q, q0)
            if accepted:
                self.accepted += 1
                l_new = lp
                self.chain_previous_lpoint[self.chain_index] = l_new
        self.steps_until_tune -= 1
        self.stage_sample += 1
        self.cumulative_samples += 1
        if self.stage_sample == self.n_steps:
            self.stage_sample = 0
        return q_new, l_new


class BatchedMetropolis(object):
    """All local chains advance together: Q [c, nparams], L [c, nllk] stay on the device.

    target      object with .nparams .nllk .batch(Q) .astep_batch(...)  (LogpForwFunc)
    per-chain scaling / acceptance counters replace the reference's per-process step state
    (metropolis.py:294-306 tuning every ``tune_interval`` steps with pymc's table)."""

    def __init__(self, target, lower, upper, n_chains, device=None, tune=True, tune_interval=100,
                 scale=1.0, seed=0):
        import torch
        self.torch = torch
        self.target = target
        self.device = device if device is not None else torch.device("cpu")
        self.lower = torch.as_tensor(np.asarray(lower, dtype=np.float64)).to(self.device)
        self.upper = torch.as_tensor(np.asarray(upper, dtype=np.float64)).to(self.device)
        self.n_chains = int(n_chains)
        self.tune, self.tune_interval = tune, int(tune_interval)
        self.scaling = torch.full((self.n_chains,), float(scale), dtype=torch.float64, device=self.device)
        self.accepted_since_tune = torch.zeros(self.n_chains, dtype=torch.int32, device=self.device)
        self.steps_until_tune = self.tune_interval
        self.n_accepted_total = 0
        self.n_steps_total = 0
        self._acc = torch.zeros(self.n_chains, dtype=torch.int32, device=self.device)
        self.seed = seed

    def set_proposal(self, cov, proposal_name="MultivariateNormal"):
        if proposal_name not in multivariate_proposals:
            raise NotImplementedError("device proposals: %s" % ", ".join(multivariate_proposals))
        self.proposal = DeviceMvNormalProposal(cov, self.device, seed=self.seed + 7919 * (1 + self.n_steps_total),
                                               df=1 if proposal_name == "MultivariateCauchy" else np.inf)

    def set_proposal_from_population(self, population, weights, proposal_name="MultivariateNormal"):
        """the same distribution as ``set_proposal(np.cov(population, aweights=weights))`` without the
        factorisation (DeviceMvNormalProposal.from_population)"""
        if proposal_name not in multivariate_proposals:
            raise NotImplementedError("device proposals: %s" % ", ".join(multivariate_proposals))
        self.proposal = DeviceMvNormalProposal.from_population(
            population, weights, self.device, seed=self.seed + 7919 * (1 + self.n_steps_total),
            df=1 if proposal_name == "MultivariateCauchy" else np.inf)

    def evaluate(self, Q):
        """stage 0 (metropolis.py:277-286): one evaluation, no move; NaN -> ValueError"""
        L = self.target.batch(Q)
        like = L[:, -1]
        if not bool(self.torch.isfinite(like).all()):
            raise ValueError("Got NaN in likelihood evaluation! Invalid model definition? "
                             "Or starting point outside prior bounds!")
        return L

    def step(self, Q, L, beta):
        """one astep for every chain; beta scalar or per-chain tensor. In place on Q, L."""
        torch = self.torch
        if self.tune and self.steps_until_tune == 0:
            acc = (self.accepted_since_tune.double() / float(self.tune_interval)).cpu().numpy()
            new = step_tune(self.scaling.cpu().numpy(), acc)
            self.scaling = torch.as_tensor(new).to(self.device)
            self.steps_until_tune = self.tune_interval
            self.accepted_since_tune.zero_()
        delta = self.proposal(self.n_chains)
        log_u = self.proposal.log_uniform(self.n_chains)
        self.target.astep_batch(Q, L, delta, self.scaling, self.lower, self.upper, log_u, beta,
                                self._acc)
        self.accepted_since_tune += self._acc
        self.steps_until_tune -= 1
        self.n_steps_total += 1
        return self._acc
