"""
``Sweeper`` with the pytensor ``Op`` call protocol of the reference
(beat/pytensorf.py:410-503): ``__props__``, ``perform(node, inputs, output)`` writing
``output[0][0]``, ``infer_shape``.  pytensor itself is not required: the class also works
eagerly (``sweeper(slownesses, nuc_dip, nuc_strike)``) and batched (``perform_batch``).
"""
import numpy as np

from .fast_sweeping import fast_sweep_ext
from .utility import positions2idxs  # noqa: F401  (re-export used next to the Sweeper)


class Sweeper(object):
    __props__ = ("patch_size", "n_patch_dip", "n_patch_strike", "implementation")

    def __init__(self, patch_size, n_patch_dip, n_patch_strike, implementation="hip"):
        self.patch_size = np.float64(patch_size)
        self.n_patch_dip = int(n_patch_dip)
        self.n_patch_strike = int(n_patch_strike)
        if implementation not in ("hip", "c"):  # "c" is accepted and served by the GPU path
            raise NotImplementedError(
                "Fast sweeping for implementation %s not implemented!" % implementation)
        self.implementation = implementation

    def perform(self, node, inputs, output):
        """pytensorf.py:443-500; inputs = (slownesses, nuc_dip, nuc_strike)"""
        slownesses, nuc_dip, nuc_strike = inputs
        z = output[0]
        z[0] = fast_sweep_ext.fast_sweep(
            np.ascontiguousarray(slownesses, dtype=np.float64), self.patch_size, int(nuc_dip),
            int(nuc_strike), self.n_patch_dip, self.n_patch_strike)

    def perform_batch(self, slownesses, nuc_dip, nuc_strike):
        """slownesses (C, n), nuc_dip/nuc_strike (C,) -> (C, n)"""
        return fast_sweep_ext.fast_sweep_batch(slownesses, self.patch_size, nuc_dip, nuc_strike,
                                               self.n_patch_dip, self.n_patch_strike)

    def __call__(self, slownesses, nuc_dip, nuc_strike):
        out = [[None]]
        self.perform(None, (slownesses, nuc_dip, nuc_strike), out)
        return out[0][0]

    def infer_shape(self, fgraph=None, node=None, input_shapes=None):
        return [(self.n_patch_dip * self.n_patch_strike,)]

    def __eq__(self, other):
        return type(self) is type(other) and all(
            getattr(self, p) == getattr(other, p) for p in self.__props__)

    def __hash__(self):
        return hash((type(self),) + tuple(getattr(self, p) for p in self.__props__))
