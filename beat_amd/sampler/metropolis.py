"""
Metropolis stepping of a whole block of chains -- counterpart of
beat/sampler/metropolis.py:276-422 (``Metropolis.astep``).

The reference advances one chain per process and call.  ``BatchedMetropolis`` keeps the state of
all chains of a rank on the device (``Q [c, nparams]``, ``L [c, nllk]``, per-chain ``scaling`` and
acceptance counters) and one ``step`` is ONE device call, ``beatamd_ffi_mstep_batch``: proposal
draws -> propose -> forward model -> tempered accept -> acceptance counters (off the device the same
in pieces: ``draw`` + ``astep_batch``) and, every ``tune_interval`` steps, the per-chain step-size
update (``beatamd_metropolis_tune``).  Nothing returns to the host inside a stage.
"""
import numpy as np

from .base import covariance_factor, proposal_df, univariate_proposals
from .ops import ops_for


class BatchedMetropolis(object):
    """target: object with .nparams .nllk .batch(Q) .astep_batch(...) (LogpForwFunc / HostTarget).
    first_chain: global index of the first local chain -- keys the random streams, so a chain
    draws the same proposals however the population is sharded over GPUs."""

    def __init__(self, target, lower, upper, n_chains, device=None, tune=True, tune_interval=100,
                 scale=1.0, seed=0, first_chain=0):
        import torch
        self.torch = torch
        self.target = target
        self.device = device if device is not None else torch.device("cpu")
        self.ops = ops_for(self.device, target)
        self.lower = torch.as_tensor(np.asarray(lower, dtype=np.float64)).to(self.device)
        self.upper = torch.as_tensor(np.asarray(upper, dtype=np.float64)).to(self.device)
        self.n_chains = int(n_chains)
        self.first_chain = int(first_chain)
        self.tune, self.tune_interval = tune, int(tune_interval)
        self.scaling = torch.full((self.n_chains,), float(scale), dtype=torch.float64, device=self.device)
        self.accepted_since_tune = torch.zeros(self.n_chains, dtype=torch.int32, device=self.device)
        self.steps_until_tune = self.tune_interval
        self.n_steps_total = 0
        self._acc = torch.zeros(self.n_chains, dtype=torch.int32, device=self.device)
        self.seed = int(seed)
        self.factor, self.df = None, 0
        self.kind, self.uscale = None, None   # per-parameter proposal family and its scales

    # -- proposal
    def _set_univariate(self, proposal_name, scale=None):
        """NormalProposal / CauchyProposal / LaplaceProposal (base.py:129-147): independent
        components; the reference's Metropolis sets scale = ones (metropolis.py:209-212)"""
        npar = int(self.lower.shape[0])
        sc = np.ones(npar) if scale is None else np.broadcast_to(np.asarray(scale, dtype=np.float64), (npar,))
        if proposal_name == "Poisson" and not (np.all(sc >= 0.0) and np.all(sc <= 500.0)):
            raise ValueError("PoissonProposal: the step widths (lam) must lie in [0, 500]")
        self.kind = univariate_proposals[proposal_name]
        self.uscale = self.torch.from_numpy(np.ascontiguousarray(sc)).to(self.device)
        self.factor, self.df = None, 0

    def set_proposal(self, cov, proposal_name="MultivariateNormal"):
        """proposal N(0, cov) / multivariate Cauchy with scale matrix cov; for the per-parameter
        families `cov` is the vector of scales (None: ones)"""
        if proposal_name in univariate_proposals:
            return self._set_univariate(proposal_name, cov)
        self.df = proposal_df(proposal_name)
        self.kind = None
        self.factor = self.torch.from_numpy(covariance_factor(cov)).to(self.device)

    def set_proposal_from_population(self, population, weights, proposal_name="MultivariateNormal"):
        """proposal with the weighted sample covariance of the population (SMC.calc_covariance,
        smc.py:167-186) through its factor; population / weights are tensors on self.device.  The
        per-parameter families keep their scales: the reference only refreshes a multivariate
        proposal between stages (smc.py:514-520)"""
        if proposal_name in univariate_proposals:
            if self.kind is None:
                self._set_univariate(proposal_name)
            return
        self.df = proposal_df(proposal_name)
        self.kind = None
        self.factor = self.ops.population_factor(population, weights)

    # -- evaluation / stepping
    def evaluate(self, Q):
        """stage 0 (metropolis.py:277-286): one evaluation, no move; NaN -> ValueError"""
        L = self.target.batch(Q)
        if not self.torch.is_tensor(L):
            L = self.torch.from_numpy(np.asarray(L))
        if not bool(self.torch.isfinite(L[:, -1]).all()):
            self.ops.check()   # an index outside the GF library raises IndexError like the reference
            raise ValueError("Got NaN in likelihood evaluation! Invalid model definition? "
                             "Or starting point outside prior bounds!")
        return L

    def _tune_if_due(self):
        if self.tune and self.steps_until_tune == 0:
            self.ops.tune(self.scaling, self.accepted_since_tune, self.tune_interval)
            self.steps_until_tune = self.tune_interval

    def run(self, Q, L, beta, n_steps, n_acc, use_graph=False):
        """n_steps steps of every chain, in place on Q, L; accepted moves are added to the 0-d int64
        tensor n_acc.  On the device a step is ONE C call (beatamd_ffi_mstep_batch: draws, proposal,
        forward model, accept, acceptance counters).  use_graph (CUDA only): a chunk of steps is captured
        ONCE per call in a HIP graph and replayed, with the Philox step counter resident on the
        device (beatamd_ctx_set_step_counter), so the draws are exactly those of the eager loop;
        the step-size tuning runs between replays."""
        torch = self.torch
        n_steps = int(n_steps)
        if not (use_graph and Q.is_cuda and n_steps >= 3 and hasattr(self.ops, "ctx")):
            for _ in range(n_steps):
                self.step(Q, L, beta, n_acc)
            return
        self.step(Q, L, beta, n_acc)      # eager: allocations, measured kernel choices
        ctx = self.ops.ctx
        counter = torch.tensor([self.n_steps_total], dtype=torch.int32, device=Q.device)
        torch.cuda.synchronize(Q.device)
        # one graph holds a CHUNK of steps (a replay costs ~10-16 us of host / front-end time whatever it contains:
        # with one 60 us step per replay the graph was slower than the eager loop it was built to beat); the step-size
        # tuning falls between chunks
        chunk = max(1, min(16, self.tune_interval if self.tune else 16, n_steps - 1))
        graph = torch.cuda.CUDAGraph()
        ctx.set_step_counter(counter)
        try:
            with torch.cuda.graph(graph):
                for _ in range(chunk):
                    self._draw_and_astep(Q, L, beta, n_acc)
            remaining = n_steps - 1
            while remaining > 0:
                self._tune_if_due()
                room = self.steps_until_tune if self.tune else remaining
                if remaining >= chunk and room >= chunk:
                    graph.replay()
                    done = chunk
                else:
                    self._draw_and_astep(Q, L, beta, n_acc)     # (eager, same device-resident step counter)
                    done = 1
                self.steps_until_tune -= done
                self.n_steps_total += done
                remaining -= done
        finally:
            ctx.set_step_counter(None)
        if int(counter.item()) != self.n_steps_total & 0x7fffffff:
            raise RuntimeError("device step counter %d != host %d" % (int(counter.item()), self.n_steps_total))

    def step(self, Q, L, beta, n_acc=None):
        """one astep for every chain; beta scalar or per-chain tensor.  In place on Q, L; the number
        of moves is added to the 0-d int64 tensor n_acc when given."""
        self._tune_if_due()
        self._draw_and_astep(Q, L, beta, n_acc)
        self.steps_until_tune -= 1
        self.n_steps_total += 1
        return self._acc

    def _draw_and_astep(self, Q, L, beta, n_acc=None):
        if self.factor is None and self.kind is None:
            raise RuntimeError("no proposal set: call set_proposal / set_proposal_from_population")
        if Q.is_cuda and hasattr(self.target, "mstep_batch"):
            # draws, proposal, forward model, accept and the acceptance counters: one device call
            self.target.mstep_batch(Q, L, self.factor if self.kind is None else self.uscale, self.kind, self.df,
                                    self.seed, self.n_steps_total, self.first_chain, self.scaling, self.lower,
                                    self.upper, beta, self._acc, self.accepted_since_tune, n_acc)
            return
        if self.kind is not None:
            delta, log_u = self.ops.draw_univariate(self.kind, self.uscale, self.n_chains, self.seed,
                                                    self.n_steps_total, first_chain=self.first_chain)
        else:
            delta, log_u = self.ops.draw(self.factor, self.n_chains, self.seed, self.n_steps_total,
                                         first_chain=self.first_chain, df=self.df)
        self.target.astep_batch(Q, L, delta, self.scaling, self.lower, self.upper, log_u, beta,
                                self._acc)
        self.accepted_since_tune += self._acc
        if n_acc is not None:
            n_acc += self._acc.sum()

    # -- resume support
    def state_dict(self):
        return dict(scaling=self.scaling.cpu().numpy(), n_steps_total=self.n_steps_total,
                    steps_until_tune=self.steps_until_tune, seed=self.seed,
                    accepted_since_tune=self.accepted_since_tune.cpu().numpy())

    def load_state_dict(self, st, block=None):
        """st holds GLOBAL per-chain arrays when block = (start, stop) is given"""
        sl = slice(*block) if block is not None else slice(None)
        self.scaling.copy_(self.torch.from_numpy(np.asarray(st["scaling"], dtype=np.float64)[sl]))
        self.accepted_since_tune.copy_(
            self.torch.from_numpy(np.asarray(st["accepted_since_tune"], dtype=np.int32)[sl]))
        self.n_steps_total = int(st["n_steps_total"])
        self.steps_until_tune = int(st["steps_until_tune"])
        if "seed" in st:   # the proposal stream is keyed by (seed, step counter, chain)
            self.seed = int(st["seed"])
