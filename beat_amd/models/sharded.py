"""
Seismic libraries SHARDED BY TARGET over the ranks ("if a library ever exceeds HBM: shard by
target -- each GPU owns T/R targets for all chains, per-chain partial sums, one small collective of (C,)" -- never by
patch).

The default multi-GPU layout replicates the libraries and shards the CHAINS (beat_amd/parallel.py).  A library on the
reference tutorial's (17 durations x 41 start times) grid with 64 targets, 400 patches and two slip components
passes 288 GB at ~1000 samples per trace; then every rank keeps the rows of its contiguous block of targets only and
evaluates ALL chains on them:

    rank r:  logpts[c, t] for its targets t  (the same kernels on a wavemap of T_r targets: stacking, residual,
             multivariate_normal_chol -- reference beat/models/seismic.py:1253-1349; every target is independent)
    all ranks: one all-gather of (T + R) x C doubles per evaluation -> the full likelihood vector of every chain in
             target order, `like` summed in the fused model's order (beat/models/problems.py:227-247: per composite,
             then over composites) -- from the SAME gathered bits on every rank, whatever the number of ranks

so the likelihood vectors, and with them every accept decision and stage transition, are bit-identical on all ranks
and to the replicated run.  The geodetic and Laplacian composites are small and stay replicated.

``TargetShardedLogp`` has the batched interface of ``LogpForwFunc`` that the samplers use (``nparams``, ``nllk``,
``batch``, ``astep_batch``, ``update_weights``, ``release``); the Metropolis step runs in pieces -- draws on the device,
``beatamd_metropolis_propose``, forward model, ONE all-gather, ``beatamd_like_assemble``, ``beatamd_metropolis_accept``:
the kernels of the fused step (round 6; torch glue before) -- because a collective sits between forward model and
acceptance.  ``SMC(..., shard="targets")`` keeps all chains on every
rank.
"""
import numpy as np

from .. import parallel
from .problem import FFIProblem, SeismicWavemap


def target_block(n_targets, rank, world):
    """contiguous block of targets owned by ``rank`` -> (start, stop); the remainder goes to the first ranks"""
    return parallel.chain_block(n_targets, rank, world)


def shard_wavemap(wm, rank, world):
    """the wavemap restricted to this rank's targets: library rows (a view of the host array / device tensor: the
    target axis is the slowest), data, weights, log-determinants, hyper-parameter and station-shift indices"""
    from ..ffi import SeismicGFLibrary, SeismicGFLibraryConfig
    a, b = target_block(wm.n_t, rank, world)
    gfs = {}
    for v, gf in wm.gfs.items():
        T, P, D, S, N = gf.dimensions
        cfg = SeismicGFLibraryConfig(dimensions=(b - a, P, D, S, N), starttime_sampling=gf.starttime_sampling,
                                     duration_sampling=gf.duration_sampling, starttime_min=gf.starttime_min,
                                     duration_min=gf.duration_min, component=getattr(gf.config, "component", v))
        part = SeismicGFLibrary(cfg)
        if getattr(gf, "_device_tensor", None) is not None:
            part.adopt_device_tensor(gf._device_tensor[a:b])
        else:
            part.setup(b - a, P, D, S, N, allocate=False)
            part._gfmatrix = gf._gfmatrix[a:b]
        gfs[v] = part
    w = wm.weights
    if hasattr(w, "is_cuda"):            # (operators left on the device by update_weights)
        w = w.detach().cpu().numpy()
    w = np.asarray(w)
    if getattr(wm, "is_prewhitened", False) or getattr(wm, "_whitened_with", None) is not None:
        raise ValueError("target sharding of a PRE-WHITENED wavemap is not supported: shard the problem first, then compile "
                         "each rank's block with prewhiten=...")
    ts = None if wm.time_shifts is None else (wm.time_shifts[0], np.asarray(wm.time_shifts[1])[a:b])
    return SeismicWavemap(gfs, wm.data[a:b], w[a:b], np.asarray(wm.slog_pdet)[a:b], wm.hypers[a:b], ts, wm.interpolation, wm.name)


def shard_problem(prob, rank, world):
    """FFIProblem whose wavemaps hold this rank's targets; geodetic data and the Laplacian stay whole"""
    return FFIProblem(prob.layout, prob.n_patch_dip, prob.n_patch_strike, prob.patch_sizes, prob.slip_varnames,
                      [shard_wavemap(wm, rank, world) for wm in prob.wavemaps], prob.geodetic, prob.laplacian,
                      prob.lower, prob.upper)


class TargetShardedLogp(object):
    """the fused log-likelihood of an FFI problem with target-sharded seismic libraries.

    prob: the WHOLE problem description (libraries may be memory-mapped / lazily loaded: only this rank's target block
    is uploaded).  Every rank constructs it with the same arguments."""

    def __init__(self, prob, ctx=None, rank=None, world=None):
        import torch
        self.torch = torch
        r, w, _ = parallel.dist_info()
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        self.problem = prob
        self.n_t = [wm.n_t for wm in prob.wavemaps]
        for n in self.n_t:
            if n < self.world:
                raise ValueError("target sharding needs at least one target per rank and wavemap (%d targets, %d ranks)"
                                 % (n, self.world))
        self.local = shard_problem(prob, self.rank, self.world).compile(ctx)
        self.ctx = self.local.ctx
        self.nparams = self.local.nparams
        self.blocks = [[target_block(n, q, self.world) for q in range(self.world)] for n in self.n_t]
        self.n_local = [b[self.rank][1] - b[self.rank][0] for b in self.blocks]
        n_seis_local = sum(self.n_local)
        self.n_rest = self.local.nllk - n_seis_local - 1         # geodetic datasets, Laplacian: replicated columns
        self.nllk = sum(self.n_t) + self.n_rest + 1
        # composite boundaries of the full vector for `like` (k_like_sum: one sum per composite, then their sum): the
        # seismic composite = all wavemaps' datasets, then the replicated composites as the local model has them
        self._seis_total = sum(self.n_t)
        self._rest_groups = self._rest_group_sizes()

    def _rest_group_sizes(self):
        g = []
        if self.problem.geodetic is not None:
            g.append(len(self.problem.geodetic.sizes))
        if self.problem.laplacian is not None:
            g.append(self.n_rest - sum(g))
        if sum(g) != self.n_rest:
            raise RuntimeError("likelihood layout of the local model not understood (%d replicated columns)" % self.n_rest)
        return g

    # -- evaluation
    def _plan(self):
        """static maps of the gathered block (the same on every rank): destination column of every gathered row (-1: a
        rank's flag row), composite boundaries of the full vector"""
        if getattr(self, "_dst_col", None) is None:
            col0 = np.concatenate([[0], np.cumsum(self.n_t)])
            dst = []
            for q in range(self.world):
                for iw in range(len(self.n_t)):
                    a, b = self.blocks[iw][q]
                    dst += [int(col0[iw]) + k for k in range(a, b)]
                dst.append(-1)
            self._dst_col = np.asarray(dst, dtype=np.int32)
            ends, e = [], self._seis_total
            ends.append(e)
            for gsz in self._rest_groups:
                e += gsz
                ends.append(e)
            self._group_end = np.asarray(ends, dtype=np.int32)
        return self._dst_col, self._group_end

    def batch(self, Q, out=None):
        """Q [C, nparams] (device tensor or numpy) -> LL [C, nllk] in the layout of the unsharded model.  Device side:
        this rank's rows transposed for the collective (one torch copy), ONE all-gather, ONE kernel (beatamd_like_assemble:
        scatter + replicated columns + `like` in k_like_sum's order + NaN flags)"""
        t = self.torch
        as_numpy = not t.is_tensor(Q)
        Qd = t.as_tensor(np.ascontiguousarray(Q)).to("cuda:%d" % self.ctx.device) if as_numpy else Q
        L_loc = self.local.batch(Qd)
        C = L_loc.shape[0]
        ns = sum(self.n_local)
        # this rank's rows: its datasets' logpts (wavemap by wavemap) + its `like` (NaN marks a chain whose times left the
        # library grid on one of ITS targets: the flag has to reach every rank)
        mine = t.empty((ns + 1, C), dtype=t.float64, device=L_loc.device)
        mine[:ns].copy_(L_loc[:, :ns].t())
        mine[ns].copy_(L_loc[:, -1])
        allr = parallel.allgather_rows(mine) if self.world > 1 else mine     # (blocks of different length: counts exchanged)
        dst_col, group_end = self._plan()
        LL = out if (out is not None and t.is_tensor(out) and out.is_contiguous()) else \
            t.empty((C, self.nllk), dtype=t.float64, device=L_loc.device)
        self.ctx.like_assemble(allr.contiguous(), dst_col, L_loc, ns, self.n_rest, self._seis_total, group_end, LL)
        self._last_bad = LL[:, -1]        # (NaN = flagged on SOME rank: the collective error check reads it)
        if out is not None and out is not LL:
            out.copy_(LL)
            return out
        return LL.cpu().numpy() if as_numpy else LL

    def astep_batch(self, Q0, L0, delta, scaling, lower, upper, log_u, beta, accepted=None):
        """metropolis.py:313-385 for all chains, in place on Q0 / L0 (device tensors): propose (k_propose), forward model +
        gather + assemble, tempered acceptance (k_accept) -- the kernels of the fused step around the collective; out-of-box
        proposals are parked on the current point and still evaluated; NaN never accepts"""
        t = self.torch
        C = Q0.shape[0]
        if getattr(self, "_qprop", None) is None or self._qprop.shape != Q0.shape:
            self._qprop = t.empty_like(Q0)
            self._inb = t.empty(C, dtype=t.int32, device=Q0.device)
            self._lprop = t.empty((C, self.nllk), dtype=t.float64, device=Q0.device)
        if accepted is None:
            accepted = t.zeros(C, dtype=t.int32, device=Q0.device)
        self.ctx.metropolis_propose(Q0, delta, scaling, lower, upper, self._qprop, self._inb)
        self.batch(self._qprop, out=self._lprop)
        self.ctx.metropolis_accept(Q0, L0, self._qprop, self._lprop, self._inb, log_u, beta, accepted)
        return accepted

    def check_collectively(self):
        """the device status word (an index outside the library, ...) is set on the rank that OWNS the offending target
        only: raised alone it would leave the other ranks waiting in the next collective.  Every rank calls
        this at the same places (SMC.select_end_points): the ranks' outcomes are all-gathered and every rank raises the
        same exception class"""
        from ..sampler.ops import collective_check
        collective_check(self.ctx, self.world)

    # -- the rest of the LogpForwFunc surface the samplers touch
    def update_weights(self, wavemap_index, weights, slog_pdet):
        """new whitening operators / scalar weights of one wavemap (all T targets given): this rank installs its block"""
        a, b = self.blocks[wavemap_index][self.rank]
        w = weights[a:b] if hasattr(weights, "__getitem__") else weights
        self.local.update_weights(wavemap_index, w, slog_pdet[a:b])

    def synthetics(self, Q, wavemap_index=0, residuals=False):
        raise NotImplementedError("synthetics of a target-sharded model: evaluate the local model (self.local.synthetics) for "
                                  "this rank's targets %s" % (self.blocks[wavemap_index][self.rank],))

    def release(self):
        self.local.release()

    def __call__(self, q):
        """B1 seam for one point: list of arrays like the compiled function's outputs"""
        ll = self.batch(np.asarray(q, dtype=np.float64)[None])[0]
        return list(ll)
