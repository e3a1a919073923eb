"""
The sampler steps that sit between forward evaluations, behind one small interface with two
back ends:

``DeviceOps``   the HIP kernels of ``csrc/smc.hip`` / ``csrc/gemm.hip`` through the C ABI
                (``beatamd_smc_calc_beta``, ``_smc_resample``, ``_smc_population_factor``,
                ``_proposal_draw``, ``_gather_rows``, ``_metropolis_tune``).  Used whenever the
                population lives on a GPU -- i.e. always for the FFI forward model.  The
                population stays in HBM across stages; only beta returns to the host.
``HostOps``     vectorised numpy on CPU tensors, for targets that are plain Python functions
                (``HostTarget``: the reference's toy problems, the gloo multi-process tests).
                The forward model never runs through it.

Reference arithmetic: beat/sampler/smc.py:133-186, 290-324 (calc_beta, weighted covariance,
resample), beat/sampler/base.py:35-71, 163-186 (proposal draws), beat/sampler/metropolis.py:294-306
(step-size tuning with pymc's table).
"""
import numpy as np

BAD_COVARIANCE = ("Sample covariances contains Inf or NaN! Please try reducing the upper and lower bounds "
                  "of hyper parameters!")
TUNE_TABLE = ((0.001, 0.1), (0.05, 0.5), (0.2, 0.9))  # acc <  x -> factor (first match)
TUNE_TABLE_UP = ((0.95, 10.0), (0.75, 2.0), (0.5, 1.1))  # acc > x -> factor (first match)


def step_tune(scale, acc_rate):
    """pymc.step_methods.metropolis.tune, as used at metropolis.py:294-306 (pymc is not in the
    reference tree; table restated from its documentation).  Vectorised over chains."""
    scale = np.asarray(scale, dtype=np.float64)
    acc = np.asarray(acc_rate, dtype=np.float64)
    f = np.ones_like(acc)
    for thr, fac in reversed(TUNE_TABLE_UP):
        f = np.where(acc > thr, fac, f)
    for thr, fac in reversed(TUNE_TABLE):
        f = np.where(acc < thr, fac, f)
    return scale * f


class HostOps(object):
    """numpy twins for CPU tensors (toy targets).  Same call signatures as DeviceOps."""

    def __init__(self):
        import torch
        self.torch = torch

    def _np(self, t):
        return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)

    def calc_beta(self, like, beta, coef_variation):
        lk = self._np(like).astype(np.float64)
        lo, hi, shifted = beta, 2.0, lk - lk.max()
        mid, temp = beta, None
        while hi - lo > 1e-6:
            mid = (lo + hi) / 2.0
            temp = np.exp((mid - beta) * shifted)
            if np.std(temp) / np.mean(temp) > coef_variation:
                hi = mid
            else:
                lo = mid
        return mid, self.torch.from_numpy(temp / np.sum(temp))

    def stage_weights(self, like, dbeta):
        lk = self._np(like).astype(np.float64)
        temp = np.exp(dbeta * (lk - lk.max()))
        return self.torch.from_numpy(temp / np.sum(temp))

    def resample(self, weights, aux):
        w = self._np(weights)
        n = w.size
        u = (np.arange(n) + float(aux)) / n
        idx = np.minimum(np.searchsorted(np.cumsum(w), u, side="left"), n - 1)
        return self.torch.from_numpy(idx.astype(np.int32))

    def population_factor(self, population, weights):
        X, w = self._np(population), self._np(weights)
        v1, v2 = w.sum(), (w * w).sum()
        mean = (w[:, None] * X).sum(0) / v1
        with np.errstate(all="ignore"):
            F = np.sqrt(w / (v1 - v2 / v1))[:, None] * (X - mean)
        if not (v1 - v2 / v1 > 0.0) or not np.isfinite(F).all():
            raise ValueError(BAD_COVARIANCE)   # smc.py:181-185
        return self.torch.from_numpy(F)

    def draw(self, factor, n_chains, seed, step, first_chain=0, df=0):
        F = self._np(factor)
        rs = np.random.RandomState((int(seed) * 1000003 + int(step) * 7919 + int(first_chain)) % (2 ** 32))
        rows = rs.standard_normal((n_chains, F.shape[0])) @ F
        if df:
            g = rs.standard_normal((n_chains, int(df)))
            rows = rows / np.sqrt((g * g).sum(1) / float(df))[:, None]
        log_u = np.log(rs.uniform(size=n_chains))
        return self.torch.from_numpy(rows), self.torch.from_numpy(log_u)

    def draw_univariate(self, kind, scale, n_chains, seed, step, first_chain=0):
        """NormalProposal / CauchyProposal / LaplaceProposal / PoissonProposal rows (base.py:129-160), numpy generator"""
        sc = self._np(scale)
        rs = np.random.RandomState((int(seed) * 1000003 + int(step) * 7919 + int(first_chain)) % (2 ** 32))
        shape = (n_chains, sc.size)
        if kind == 0:
            rows = rs.standard_normal(shape)
        elif kind == 1:
            rows = rs.standard_cauchy(shape)
        elif kind == 2:
            rows = rs.standard_exponential(shape) - rs.standard_exponential(shape)
        else:   # base.py:150-155: poisson(lam=scale) - scale
            rows = rs.poisson(lam=np.broadcast_to(sc, shape)).astype(np.float64) - sc
            log_u = np.log(rs.uniform(size=n_chains))
            return self.torch.from_numpy(rows), self.torch.from_numpy(log_u)
        log_u = np.log(rs.uniform(size=n_chains))
        return self.torch.from_numpy(rows * sc), self.torch.from_numpy(log_u)

    def gather(self, src, idx):
        return src[idx.long()].contiguous()

    def tune(self, scaling, accepted, interval):
        new = step_tune(scaling.numpy(), accepted.numpy() / float(interval))
        scaling.copy_(self.torch.from_numpy(new))
        accepted.zero_()

    def check(self):
        pass


class DeviceOps(object):
    """HIP kernels on CUDA tensors through ``beat_amd.engine.Context``."""

    def __init__(self, ctx):
        self.ctx = ctx

    def calc_beta(self, like, beta, coef_variation):
        """like: 1-D view (may be a strided column of L) -> (beta_new, weights)"""
        stride = int(like.stride(0)) if like.dim() == 1 else 1
        return self.ctx.smc_calc_beta(_Base(like), beta, coef_variation, stride=stride, n=like.shape[0])

    def stage_weights(self, like, dbeta):
        stride = int(like.stride(0)) if like.dim() == 1 else 1
        return self.ctx.smc_stage_weights(_Base(like), dbeta, stride=stride, n=like.shape[0])

    def resample(self, weights, aux):
        return self.ctx.smc_resample(weights, aux)

    def population_factor(self, population, weights):
        """tall factor F (chains, nparams) of the weighted sample covariance; with (many) more chains
        than parameters its compact Cholesky form R (nparams, nparams), R^T R = F^T F -- nparams
        instead of `chains` normals per proposal row.  A population whose covariance is not
        numerically positive definite keeps the tall factor."""
        F = self.ctx.smc_population_factor(population, weights)
        # degenerate weights / non-finite population: the kernel raised the status word; the
        # reference's calc_covariance aborts here (ValueError, smc.py:181-185) -- once per stage
        self.ctx.synchronize()
        if F.shape[0] >= 2 * F.shape[1]:
            R = self.ctx.factor_compact(F)
            if R is not None:
                return R
        return F

    def draw(self, factor, n_chains, seed, step, first_chain=0, df=0):
        return self.ctx.proposal_draw(factor, n_chains, seed, step, first_chain=first_chain, df=df)

    def draw_univariate(self, kind, scale, n_chains, seed, step, first_chain=0):
        return self.ctx.proposal_draw_univariate(kind, scale, n_chains, seed, step, first_chain=first_chain)

    def gather(self, src, idx):
        return self.ctx.gather_rows(src, idx)

    def tune(self, scaling, accepted, interval):
        self.ctx.metropolis_tune(scaling, accepted, interval)

    def check(self):
        """synchronise and raise what the reference raises for an index outside the GF library
        (IndexError) -- the device marks such chains NaN meanwhile, so they are never accepted"""
        self.ctx.synchronize()


def collective_check(ctx, world):
    """ops.check() of models whose ranks can fail alone (target-sharded libraries): every rank synchronises, the outcomes
    (0 ok, 1 IndexError, 2 ValueError, 3 LinAlgError, 4 anything else) are all-gathered and EVERY rank raises -- its own
    exception when it has one, else the class the failing rank raised (a rank raising alone leaves the
    others blocked in the next collective)"""
    import torch

    from .. import parallel
    err, code = None, 0
    try:
        ctx.synchronize()
    except IndexError as exc:
        err, code = exc, 1
    except np.linalg.LinAlgError as exc:
        err, code = exc, 3
    except ValueError as exc:
        err, code = exc, 2
    except Exception as exc:       # noqa: BLE001 (re-raised below, on every rank)
        err, code = exc, 4
    codes = [code]
    if world > 1:
        mine = torch.tensor([[float(code)]], dtype=torch.float64, device="cuda:%d" % ctx.device)
        codes = [int(x) for x in parallel.allgather_rows(mine)[:, 0].cpu().tolist()]
    if err is not None:
        raise err
    worst = max(codes)
    if worst:
        who = [r for r, c_ in enumerate(codes) if c_]
        msg = "raised on rank(s) %s (this rank's targets are fine)" % who
        raise {1: IndexError("index out of bounds of the GF library (duration/starttime outside the library grid): " + msg),
               2: ValueError(msg), 3: np.linalg.LinAlgError(msg)}.get(worst, RuntimeError(msg))


class _Base(object):
    """a strided device view handed to the C ABI as (base pointer, stride): quacks like the
    contiguous tensors ``engine.f64`` accepts"""

    def __init__(self, t):
        import torch
        if t.dtype != torch.float64:
            raise ValueError("likelihoods must be float64")
        self._t = t
        self.is_cuda = t.is_cuda
        self.device = t.device
        self.dtype = t.dtype

    def data_ptr(self):
        return self._t.data_ptr()

    def is_contiguous(self):
        return True

    def numel(self):
        return self._t.shape[0]


def ops_for(tensor_or_device, target=None):
    """DeviceOps for CUDA (needs the context of the target, or the process context of that
    device), HostOps for CPU tensors"""
    dev = getattr(tensor_or_device, "device", tensor_or_device)
    if getattr(dev, "type", "cpu") == "cuda":
        ctx = getattr(target, "ctx", None)
        if ctx is None:
            from ..engine import get_context
            ctx = get_context(dev.index or 0)
        return DeviceOps(ctx)
    return HostOps()
