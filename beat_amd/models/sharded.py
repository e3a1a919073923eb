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
``batch``, ``astep_batch``); the Metropolis step runs in pieces (draws on the device, propose, forward + gather, accept)
because a collective sits between forward model and acceptance.  ``SMC(..., shard="targets")`` keeps all chains on every
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
    w = np.asarray(wm.weights)
    ts = None if wm.time_shifts is None else (wm.time_shifts[0], np.asarray(wm.time_shifts[1])[a:b])
    return SeismicWavemap(gfs, wm.data[a:b], w[a:b], wm.slog_pdet[a:b], wm.hypers[a:b], ts, wm.interpolation, wm.name)


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
    def batch(self, Q, out=None):
        """Q [C, nparams] (device tensor or numpy) -> LL [C, nllk] in the layout of the unsharded model"""
        t = self.torch
        as_numpy = not t.is_tensor(Q)
        Qd = t.as_tensor(np.ascontiguousarray(Q)).to("cuda:%d" % self.ctx.device) if as_numpy else Q
        L_loc = self.local.batch(Qd)
        C = L_loc.shape[0]
        ns = sum(self.n_local)
        # this rank's rows: its datasets' logpts (wavemap by wavemap) + its `like` (NaN marks a chain whose times left the
        # library grid on one of ITS targets: the flag has to reach every rank)
        mine = t.cat([L_loc[:, :ns].t(), L_loc[:, -1:].t()], 0).contiguous()
        allr = parallel.allgather_rows(mine) if self.world > 1 else mine     # (blocks of different length: counts exchanged)
        LL = t.empty((C, self.nllk), dtype=t.float64, device=L_loc.device)
        bad = t.zeros(C, dtype=t.bool, device=L_loc.device)
        col0 = 0
        rank_rows = [sum(b[q][1] - b[q][0] for b in self.blocks) + 1 for q in range(self.world)]
        starts = np.concatenate([[0], np.cumsum(rank_rows)])
        for iw, n in enumerate(self.n_t):
            for q in range(self.world):
                a, b = self.blocks[iw][q]
                before = sum(self.blocks[k][q][1] - self.blocks[k][q][0] for k in range(iw))
                r0 = int(starts[q]) + before
                LL[:, col0 + a:col0 + b] = allr[r0:r0 + (b - a)].t()
            col0 += n
        for q in range(self.world):
            bad |= t.isnan(allr[int(starts[q + 1]) - 1])
        LL[:, col0:col0 + self.n_rest] = L_loc[:, ns:ns + self.n_rest]
        # like: the sums of k_like_sum (beat_amd/csrc/logp.hip; problems.py:227-247), term by term in column order
        total = t.zeros(C, dtype=t.float64, device=L_loc.device)
        k = 0
        for gsz in [self._seis_total] + self._rest_groups:
            s = t.zeros(C, dtype=t.float64, device=L_loc.device)
            for _ in range(gsz):
                s = s + LL[:, k]
                k += 1
            total = total + s
        LL[:, -1] = t.where(bad, t.full_like(total, float("nan")), total)
        if out is not None:
            out.copy_(LL)
            return out
        return LL.cpu().numpy() if as_numpy else LL

    def astep_batch(self, Q0, L0, delta, scaling, lower, upper, log_u, beta, accepted=None):
        """metropolis.py:313-385 for all chains, in place on Q0 / L0 (device tensors): propose, prior box, forward
        model + gather, tempered acceptance -- the decisions of the fused kernels (out-of-box proposals are parked on
        the current point and still evaluated; NaN never accepts)"""
        t = self.torch
        q = Q0 + delta * scaling[:, None]
        inb = ((q >= lower) & (q <= upper)).all(1)
        qe = t.where(inb[:, None], q, Q0)
        lp = self.batch(qe)
        b = beta if t.is_tensor(beta) and beta.ndim else float(beta)
        mr = b * (lp[:, -1] - L0[:, -1])
        acc = inb & t.isfinite(mr) & (log_u < mr)
        Q0.copy_(t.where(acc[:, None], q, Q0))
        L0.copy_(t.where(acc[:, None], lp, L0))
        if accepted is None:
            accepted = t.zeros(Q0.shape[0], dtype=t.int32, device=Q0.device)
        accepted.copy_(acc.to(t.int32))
        return accepted

    def __call__(self, q):
        """B1 seam for one point: list of arrays like the compiled function's outputs"""
        ll = self.batch(np.asarray(q, dtype=np.float64)[None])[0]
        return list(ll)
