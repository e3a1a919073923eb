"""A log-likelihood given as a plain vectorised Python function, with the same batched
interface as the compiled FFI model.  For user-defined / toy targets (the reference's
test_smc.py two-Gaussian problem); the FFI forward model never goes through this class."""
import numpy as np


class HostTarget(object):
    def __init__(self, fn, nparams):
        """fn(Q (c, nparams) ndarray) -> like (c,)"""
        self.fn, self.nparams, self.nllk = fn, int(nparams), 1

    def batch(self, Q, out=None):
        import torch
        q = Q.detach().cpu().numpy() if hasattr(Q, "detach") else np.asarray(Q)
        ll = np.asarray(self.fn(q), dtype=np.float64).reshape(-1, 1)
        return torch.from_numpy(ll) if hasattr(Q, "detach") else ll

    def astep_batch(self, Q0, L0, delta, scaling, lower, upper, log_u, beta, accepted=None):
        """metropolis.py:313-385 vectorised over chains (torch CPU tensors, in place)"""
        import torch
        q = Q0 + delta * scaling[:, None]
        inb = ((q >= lower) & (q <= upper)).all(1)
        qe = torch.where(inb[:, None], q, Q0)
        lp = self.batch(qe)
        b = beta if hasattr(beta, "shape") and getattr(beta, "ndim", 0) else float(beta)
        mr = b * (lp[:, -1] - L0[:, -1])
        acc = inb & torch.isfinite(mr) & (log_u < mr)
        Q0[acc] = q[acc]
        L0[acc] = lp[acc]
        if accepted is None:
            accepted = torch.zeros(Q0.shape[0], dtype=torch.int32)
        accepted.copy_(acc.to(torch.int32))
        return accepted
