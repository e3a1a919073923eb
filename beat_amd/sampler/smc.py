"""
Sequential Monte Carlo (CATMIP / TMCMC) -- counterpart of beat/sampler/smc.py.

Stage logic of the reference (``calc_beta`` :133-165, ``calc_covariance`` :167-186,
``select_end_points`` :188-240, ``resample`` :290-324, ``smc_sample`` :333-546) with a
different data flow:

  * all chains of a rank advance together on the GPU (``BatchedMetropolis``): the reference's
    fork pool over chains (sampler/base.py:428-595) is the batch dimension of the kernels;
  * chains are sharded over ranks in contiguous blocks; at a stage boundary the end points and
    likelihood vectors are all-gathered (RCCL over xGMI) and stay ON THE DEVICE: the tempering
    step, the importance weights, the resampling indices, the proposal factor and the restart
    points are computed there by the kernels of ``csrc/smc.hip`` -- identically on every rank
    (fixed-order reductions, one shared auxiliary draw), so there is no rank-0 bottleneck, no
    files and no host copy of the population.  Only ``beta`` (8 bytes) returns per stage.

``array_population`` / ``array_lpoints`` / ``weights`` / ``resampling_indexes`` /
``likelihoods`` keep the reference's attribute names as host views for traces and inspection.
"""
import logging
import os

import numpy as np

from .. import parallel
from .base import proposal_df
from .metropolis import BatchedMetropolis
from .ops import ops_for

logger = logging.getLogger("smc")

sample_factor_final_stage = 1  # smc.py:24


class SMC(object):
    """Stage state of the sampler; attribute names follow the reference class."""

    def __init__(self, target, lower, upper, n_chains=100, tune=True, tune_interval=100,
                 coef_variation=1.0, check_bound=True, proposal_name="MultivariateNormal",
                 device=None, random_seed=42, scale=1.0, use_graph=False, shard="chains"):
        import torch
        self.torch = torch
        proposal_df(proposal_name)  # validates the name
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
        self.rng = np.random.RandomState(random_seed)  # identical on every rank
        self.rank, self.world, _ = parallel.ensure_group()
        # shard = "chains" (default): every rank steps its contiguous block of chains on a replicated model, the end
        # points are all-gathered per stage.  "targets": the MODEL is sharded (beat_amd.models.sharded: each rank holds
        # the library rows of its targets) -- every rank steps ALL chains with identical decisions, nothing to gather
        if shard not in ("chains", "targets"):
            raise ValueError("shard must be 'chains' or 'targets'")
        self.shard = shard
        self.block = (0, self.n_chains) if shard == "targets" else parallel.chain_block(self.n_chains, self.rank, self.world)
        n_local = self.block[1] - self.block[0]
        self.stepper = BatchedMetropolis(target, lower, upper, n_local, device=device, tune=tune,
                                         tune_interval=tune_interval, scale=scale, seed=random_seed,
                                         first_chain=self.block[0])
        self.device = self.stepper.device
        self.ops = ops_for(self.device, target)
        self.n_steps = 1
        # replay the Metropolis step of a stage from a HIP graph (launch-bound problems: geometry mode)
        self.use_graph = bool(use_graph)
        self.stage_betas, self.stage_acceptance = [], []
        # gathered population of all ranks (tensors on self.device), weights, restart indices
        self.Q_all = self.L_all = self.w = self.idx = None
        self.covariance = None  # kept for API parity; the proposal uses the population factor

    # ------------------------------------------------------------------ host views
    @staticmethod
    def _host(t):
        return None if t is None else t.detach().cpu().numpy()

    array_population = property(lambda self: self._host(self.Q_all))
    array_lpoints = property(lambda self: self._host(self.L_all))
    weights = property(lambda self: self._host(self.w))
    resampling_indexes = property(
        lambda self: np.arange(self.n_chains) if self.idx is None else self._host(self.idx).astype(np.int64))

    @property
    def likelihoods(self):
        return None if self.L_all is None else self._host(self.L_all[:, -1])

    # ------------------------------------------------------------------ population
    def initialize_population(self):
        """metropolis.py:125-152: prior draws (Uniform boxes) for every chain; identical on
        every rank because the seeded RandomState is shared.  -> this rank's block on the device"""
        u = self.rng.random_sample((self.n_chains, self.lower.size))
        pop = self.lower + (self.upper - self.lower) * u
        a, b = self.block
        return self.torch.from_numpy(np.ascontiguousarray(pop[a:b])).to(self.device)

    def select_end_points(self, Q_local, L_local):
        """smc.py:188-240 -- instead of reading trace files: all-gather the ranks' blocks; the
        gathered arrays stay on the device."""
        if self.shard == "targets" and hasattr(self.target, "check_collectively"):
            self.target.check_collectively()     # (a rank that owns the offending target must not raise alone: ADVICE r5)
        else:
            self.ops.check()  # surfaces an out-of-library index of the finished stage (IndexError)
        if self.shard == "targets":
            self.Q_all, self.L_all = Q_local, L_local      # every rank stepped the whole population
        else:
            self.Q_all, self.L_all = parallel.allgather_population(Q_local, L_local, self.n_chains)
        if self.Q_all.shape[0] != self.n_chains:
            raise RuntimeError("gathered %d chains, expected %d (process group not initialised?)"
                               % (self.Q_all.shape[0], self.n_chains))
        return self.Q_all, self.L_all

    def get_map_end_points(self):
        """smc.py:277-288"""
        return self._host(self.Q_all[int(self.torch.argmax(self.L_all[:, -1]))])

    # ------------------------------------------------------------------ stage transition
    def calc_beta(self):
        """smc.py:133-165 on the gathered likelihoods -> (beta, old_beta, weights tensor)"""
        beta_new, w = self.ops.calc_beta(self.L_all[:, -1], self.beta, self.coef_variation)
        return beta_new, self.beta, w

    def resample(self):
        """smc.py:290-324; the single auxiliary draw comes from the rank-shared RandomState"""
        aux = float(self.rng.rand(1)[0])
        return self.ops.resample(self.w, aux)

    def calc_covariance(self):
        """smc.py:167-186, for inspection only (host): the samplers never form it"""
        F = self._host(self.ops.population_factor(self.Q_all, self.w))
        return F.T @ F

    def transition(self, final=False):
        """everything between two stages, on the device: weights (and the next beta), proposal
        factor of the weighted population, resampling indices.  Returns False when beta passed 1
        (the caller then runs the final stage)."""
        if final:
            # smc.py:526-529: weights towards beta = 1 from the beta the population was sampled at
            self.w = self.ops.stage_weights(self.L_all[:, -1], 1.0 - self.beta)
            self.old_beta, self.beta = self.beta, 1.0
        else:
            beta_new, old, w = self.calc_beta()
            if beta_new > 1.0:
                return False
            self.beta, self.old_beta, self.w = beta_new, old, w
        self.stepper.set_proposal_from_population(self.Q_all, self.w, self.proposal_name)
        self.idx = self.resample()
        if os.environ.get("BEATAMD_CHECK_RANKS"):
            fac = self.stepper.factor if self.stepper.factor is not None else self.stepper.uscale
            parallel.assert_same_on_all_ranks("SMC stage %d transition (beta, weights, indices, proposal factor)"
                                              % self.stage, self.torch.tensor([self.beta], dtype=self.torch.float64,
                                                                              device=self.w.device),
                                              self.w, self.idx, fac)
        return True

    def restart_points(self):
        """every local chain starts at its resampled parent (sampler/base.py:541-571)"""
        a, b = self.block
        sel = self.idx[a:b].contiguous()
        return self.ops.gather(self.Q_all, sel), self.ops.gather(self.L_all, sel)

    def sample_stage(self, n_steps, on_step=None, buffer_thinning=None):
        """iter_parallel_chains for one stage: n_steps Metropolis steps of every local chain at
        self.beta.  ``buffer_thinning`` (the reference's trace option, beat/backend.py:100-117, 365-404): the states of
        every chain after the draws ``backend.thinned_draws(n_steps, thinning)`` -- every thinning-th and the last --
        are kept ON THE DEVICE (self.stage_draws = (Q [draws, chains, nparams], L [draws, chains, nllk]), this rank's
        chains) for the stage's trace files; None: end points only"""
        Q, L = self.restart_points()
        n_acc = self.torch.zeros((), dtype=self.torch.int64, device=Q.device)
        self.stage_draws = None
        if buffer_thinning is not None and on_step is None:
            from ..backend import thinned_draws
            keep = thinned_draws(n_steps, buffer_thinning)
            DQ = self.torch.empty((len(keep),) + tuple(Q.shape), dtype=Q.dtype, device=Q.device)
            DL = self.torch.empty((len(keep),) + tuple(L.shape), dtype=L.dtype, device=L.device)
            prev = -1
            for j, i in enumerate(keep):
                self.stepper.run(Q, L, self.beta, i - prev, n_acc, use_graph=self.use_graph)
                DQ[j].copy_(Q)
                DL[j].copy_(L)
                prev = i
            self.stage_draws = (DQ, DL)
        elif on_step is None:
            self.stepper.run(Q, L, self.beta, n_steps, n_acc, use_graph=self.use_graph)
        else:
            for i in range(int(n_steps)):
                acc = self.stepper.step(Q, L, self.beta, n_acc)
                on_step(i, Q, L, acc)
        self.stage_acceptance.append(float(n_acc.item()) / max(1.0, float(n_steps) * Q.shape[0]))
        return Q, L


def _dump_stage(step, homepath, layout, out_names, backend):
    """stage directory with one-draw traces of every chain (rank 0 writes; every rank holds the
    same gathered arrays) + the sampler state needed to resume (smc.py:549-557)"""
    if homepath is None:
        return
    step._stage_files_on = True
    # per-chain step state of all ranks (scaling, acceptance counters) for an exact resume
    st = step.stepper.state_dict()
    if getattr(step, "shard", "chains") == "targets":     # every rank holds every chain's step state
        sc, ac = step.stepper.scaling.cpu().numpy(), step.stepper.accepted_since_tune.cpu().numpy()
    else:
        sc = parallel.allgather_rows(step.stepper.scaling[:, None])[:, 0].cpu().numpy()
        ac = parallel.allgather_rows(step.stepper.accepted_since_tune[:, None])[:, 0].cpu().numpy()
    _join_stage_writer(step)          # (at most one stage is being written at a time)
    draws = getattr(step, "stage_draws", None)
    if draws is not None and layout is not None and step.stage != 0:
        # every kept draw of every chain (smc_sample(buffer_thinning=...)): each rank writes the trace files of ITS chains
        # -- like the reference's workers (sampler/base.py:316-395) --, in line (a stage's draws are (draws x chains x
        # (nparams + nllk)) doubles: they leave the device once, here)
        from ..backend import write_population
        a_ = 0 if getattr(step, "shard", "chains") == "targets" else step.block[0]
        if getattr(step, "shard", "chains") != "targets" or step.rank == 0:
            write_population(homepath, step.stage, layout, out_names, draws[0].cpu().numpy(), draws[1].cpu().numpy(), backend,
                             first_chain=a_)
        step.stage_draws = None
        layout = None                 # (rank 0 below: the state archive only)
    if step.rank != 0:
        if not getattr(step, "async_stage_files", True):
            _join_stage_writer(step)  # (in-line writing: rank 0's outcome is broadcast right behind its write, below)
        return
    # everything the files hold is copied to the host HERE; the writing itself (one NumpyChain / TextChain file per
    # chain + the state archive: ~0.5 ms per chain) runs in a thread beside the next stage's sampling, whose host
    # side sits in GIL-free C calls
    pop, lp = np.array(step.array_population, copy=True), np.array(step.array_lpoints, copy=True)   # (host tensors are views)
    sc, ac = np.array(sc, copy=True), np.array(ac, copy=True)
    rs = step.rng.get_state()
    state = dict(beta=step.beta, old_beta=step.old_beta, stage=step.stage, population=pop, lpoints=lp, scaling=sc,
                 accepted_since_tune=ac, n_steps_total=st["n_steps_total"], steps_until_tune=st["steps_until_tune"],
                 seed=st["seed"], rng_keys=rs[1], rng_pos=rs[2], rng_has_gauss=rs[3], rng_cached=rs[4])
    if getattr(step, "update_map_point", None) is not None:
        # the point the weights of this stage were estimated at (smc.py:492-503): a resumed run re-derives the
        # weights from it -- the saved lpoints belong to THOSE weights (the reference pickles update.get_weights()
        # in save_sampler_state; 8.6 GB of operators per wavemap here, a parameter vector does the same)
        state["update_map_point"] = np.asarray(step.update_map_point, dtype=np.float64)
    stage = step.stage

    def write():
        from ..backend import stage_path, write_population
        path = write_population(homepath, stage, layout, out_names, pop, lp, backend) if layout is not None else None
        if path is None:
            path = stage_path(homepath, stage)
            os.makedirs(path, exist_ok=True)
        # (atomic: a reader -- an on_stage callback, a resume -- sees the complete archive or none; np.savez keeps the
        # name it is given when it ends in .npz)
        tmp = os.path.join(path, ".sampler_state.tmp.npz")
        np.savez(tmp, **state)
        os.replace(tmp, os.path.join(path, "sampler_state.npz"))

    box = {}
    if not getattr(step, "async_stage_files", True):
        # in line -- but a failure is surfaced where the threaded writer's is: by _join_stage_writer, on EVERY rank (rank 0
        # raising here alone would leave the others in the next collective)
        try:
            write()
        except BaseException as exc:      # noqa: BLE001
            box["error"] = exc
        step._stage_writer = (None, box)
        return _join_stage_writer(step)
    import threading

    def guarded():
        try:
            write()
        except BaseException as exc:      # re-raised in the sampling thread by _join_stage_writer
            box["error"] = exc
    th = threading.Thread(target=guarded, name="beatamd-stage-writer", daemon=False)
    th.start()
    step._stage_writer = (th, box)


def _join_stage_writer(step):
    """wait for the stage files in flight (before the next ones are written, before an on_stage callback or a resume
    reads them, at the end of smc_sample) and surface a failure of the writer ON EVERY RANK: only rank 0 writes, and a
    rank that raised alone would leave the others waiting in the next collective.  Called by all ranks at
    the same places; with several ranks that costs one broadcast of a flag per stage."""
    w = getattr(step, "_stage_writer", None)
    err = None
    if w is not None:
        step._stage_writer = None
        if w[0] is not None:
            w[0].join()
        err = w[1].get("error")
    if getattr(step, "world", 1) > 1 and getattr(step, "_stage_files_on", False):
        failed = parallel.broadcast_array(np.array([1.0 if err is not None else 0.0]), src=0)
        if failed[0] and err is None:
            err = RuntimeError("rank 0 failed to write the stage files")
    if err is not None:
        raise err


def load_stage(step, homepath, stage):
    """resume: restore the population, the tempering state, the per-chain step sizes and the
    shared random stream of a completed stage (init_stage / load_sampler_params,
    sampler/base.py:618-661, backend.py:1049-1071)"""
    from ..backend import stage_path
    z = np.load(os.path.join(stage_path(homepath, stage), "sampler_state.npz"))
    step.beta, step.old_beta, step.stage = float(z["beta"]), float(z["old_beta"]), int(z["stage"])
    t = step.torch
    if z["population"].shape[0] != step.n_chains:
        raise ValueError("stage %s holds %d chains, the sampler was set up for %d: resume with the same n_chains"
                         % (stage, z["population"].shape[0], step.n_chains))
    step.Q_all = t.from_numpy(np.ascontiguousarray(z["population"])).to(step.device)
    step.L_all = t.from_numpy(np.ascontiguousarray(z["lpoints"])).to(step.device)
    if "scaling" in z.files:
        if z["scaling"].size != step.n_chains:
            raise ValueError("stage %s holds the step sizes of %d chains, expected %d"
                             % (stage, z["scaling"].size, step.n_chains))
        # (the step counter and the seed key the proposal streams: a resumed run must not reuse them)
        step.stepper.load_state_dict(dict(scaling=z["scaling"], accepted_since_tune=z["accepted_since_tune"],
                                          n_steps_total=z["n_steps_total"],
                                          steps_until_tune=z["steps_until_tune"],
                                          **({"seed": z["seed"]} if "seed" in z.files else {})),
                                     block=step.block)
    if "rng_keys" in z.files:
        step.rng.set_state(("MT19937", z["rng_keys"], int(z["rng_pos"]), int(z["rng_has_gauss"]),
                            float(z["rng_cached"])))
    step.update_map_point = np.array(z["update_map_point"]) if "update_map_point" in z.files else None
    return step


def _update_covariances(step, update, Q_local):
    """smc.py:492-503: new weights from the maximum-likelihood end point, then the end points evaluated again"""
    logger.info("Updating Covariances ...")
    step.update_map_point = step.get_map_end_points()
    update.update_weights(step.update_map_point)
    return update_last_samples(step, Q_local)


def update_last_samples(step, Q_local):
    """sampler/base.py:664-704: the end points of the stage are evaluated again with the updated
    weights (one draw at stage 0 = evaluation without a move, metropolis.py:277-286) and gathered"""
    L = step.stepper.evaluate(Q_local)
    return step.select_end_points(Q_local, L)


def smc_sample(n_steps, step, progressbar=False, on_stage=None, max_stages=200, homepath=None,
               layout=None, out_names=None, backend="bin", resume_stage=None, update=None, final_stage=True,
               async_stage_files=True, buffer_thinning=None):
    """smc.py:333-546 stage loop.  Returns the final population (n_chains, nparams), the
    likelihood vectors (host arrays) and the list of betas.  With ``homepath`` every stage leaves
    a ``stage_<k>`` / ``stage_final`` directory of NumpyChain/TextChain traces
    (beat_amd.backend) and the state to resume from.

    ``update`` (smc.py:492-503, ``update_covariances`` of the reference's config): an object with
    ``update_weights(map_point)`` -- e.g. ``beat_amd.covariance.NoiseCovarianceUpdate`` -- called with
    the maximum-likelihood end point after every stage; the population is then evaluated again with
    the new weights before the next tempering step is chosen.  Every rank holds the same gathered
    population, so every rank updates its own model copy identically.

    ``async_stage_files`` (default on): the stage directories are written by a thread beside the next stage's sampling;
    the writer is joined before the next stage's files, before every ``on_stage`` callback (which may read the
    directory) and at the end of the call; ``sampler_state.npz`` appears atomically.  Off: written in line.

    ``buffer_thinning`` (needs ``homepath`` and ``layout``): the stage directories hold every ``buffer_thinning``-th draw
    of every chain and the last one -- the reference's traces (``buffer_thinning`` of its sampler config,
    beat/backend.py:100-117, 365-404; ``beat summarize`` reads the whole stage) -- instead of the end points only
    (= ``buffer_thinning >= n_steps``).  Each rank writes the files of its own chains.

    ``final_stage=False`` stops after ``max_stages`` tempering stages WITHOUT the stage at beta = 1 (the state of the
    last stage can be resumed from its directory); by default a run that exhausts ``max_stages`` still ends with the
    final stage like a converged one."""
    import time
    step.n_steps = int(n_steps)
    step.async_stage_files = bool(async_stage_files)
    if n_steps < 1:
        raise TypeError("Argument `n_steps` should be above 0.")
    step.update_map_point = None
    thin = buffer_thinning if (buffer_thinning is not None and homepath is not None and layout is not None) else None
    if buffer_thinning is not None and thin is None:
        raise ValueError("buffer_thinning writes trace files: give homepath and layout (and out_names)")
    # wall-clock split of the call (every part ends on a host-visible result, so no extra synchronisation):
    # Metropolis steps | stage transitions (weights, beta, proposal factor, resampling, restart gathers) |
    # all-gather of the end points | covariance updates | trace / state files
    tm = step.timings = dict(sample_s=0.0, transition_s=0.0, gather_s=0.0, update_s=0.0, io_s=0.0, steps=0)

    def timed(key, fn, *a, **kw):
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        tm[key] += time.perf_counter() - t0
        return out

    if resume_stage is not None:
        load_stage(step, homepath, resume_stage)
        if update is not None and step.update_map_point is not None:
            # the saved likelihoods were evaluated with the weights of the stage's covariance update: install
            # them again before anything is proposed against those likelihoods
            update.update_weights(step.update_map_point)
    else:
        # stage 0: evaluate the prior population (draws = 1, no move)
        Q = step.initialize_population()
        L = timed("sample_s", step.stepper.evaluate, Q)
        timed("gather_s", step.select_end_points, Q, L)
        if update is not None:
            # the reference's update block sits inside the stage loop and therefore also runs after the initial
            # stage, BEFORE the first calc_beta (smc.py:459-503)
            timed("update_s", _update_covariances, step, update, Q)
        timed("io_s", _dump_stage, step, homepath, layout, out_names, backend)
    betas = [step.beta]

    def synced_transition(**kw):
        ok = step.transition(**kw)
        if ok and step.idx is not None and hasattr(step.idx, "is_cuda") and step.idx.is_cuda:
            step.torch.cuda.synchronize(step.idx.device)    # (the split below is wall clock)
        return ok

    while step.beta < 1.0 and step.stage < max_stages:
        if not timed("transition_s", synced_transition):
            break
        step.stage += 1
        logger.info("Beta: %f Stage: %i", step.beta, step.stage)
        Q, L = timed("sample_s", step.sample_stage, n_steps, buffer_thinning=thin)
        tm["steps"] += int(n_steps)
        timed("gather_s", step.select_end_points, Q, L)
        if update is not None:
            timed("update_s", _update_covariances, step, update, Q)
        betas.append(step.beta)
        timed("io_s", _dump_stage, step, homepath, layout, out_names, backend)
        if on_stage is not None:
            if homepath is not None:
                timed("io_s", _join_stage_writer, step)     # (the callback may read the stage directory)
            on_stage(step)
    if not final_stage and step.beta < 1.0:
        step.stage_betas = betas
        timed("io_s", _join_stage_writer, step)
        return step.array_population, step.array_lpoints, betas
    # final stage at beta = 1 (smc.py:526-543)
    step.stage = -1
    timed("transition_s", synced_transition, final=True)
    Q, L = timed("sample_s", step.sample_stage, n_steps * sample_factor_final_stage, buffer_thinning=thin)
    tm["steps"] += int(n_steps * sample_factor_final_stage)
    timed("gather_s", step.select_end_points, Q, L)
    betas.append(1.0)
    timed("io_s", _dump_stage, step, homepath, layout, out_names, backend)
    timed("io_s", _join_stage_writer, step)
    step.stage_betas = betas
    return step.array_population, step.array_lpoints, betas


def tune(acc_rate):
    """smc.py:558-575 (Muto & Beck 2008)"""
    a, b = 1.0 / 9, 8.0 / 9
    return np.power((a + (b * acc_rate)), 2)
