"""
``Sweeper``, ``GeoSynthesizer`` and ``SeisSynthesizer`` with the pytensor ``Op`` call protocol of the reference
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


# pytensorf.py:25-126 / :129-311: the geometry-mode Ops.  Same constructor arguments, ``__props__``,
# ``perform(node, inputs, output)`` and ``infer_shape``; the inputs arrive in the order of
# ``varnames`` (make_node records the keys of the input dict, pytensorf.py:80-82).
km_vars = ("east_shift", "north_shift", "depth", "length", "width")   # utility.py:30-55 kmtypes


def _update_sources(sources, mapping, varnames, inputs):
    """utility.adjust_point_units (km -> m, utility.py:651-675), split_point (:678-770: value i of
    a variable goes to the i-th source the mapping lists for it) and update_source (:773-800)"""
    for name, values in zip(varnames, inputs):
        values = np.atleast_1d(np.asarray(values, dtype=np.float64))
        if name in km_vars:
            values = values * 1000.0
        idxs = mapping[name] if mapping is not None and name in mapping else range(len(sources))
        for v, i_s in zip(values, idxs):
            sources[i_s].update(**{name: v})
    for s in sources:
        s.time = 0.0   # pytensorf.py:112-113


class GeoSynthesizer(object):
    __props__ = ("engine", "sources", "targets", "mapping")

    def __init__(self, engine, sources, targets, mapping=None):
        self.engine = engine
        self.sources = tuple(sources)
        self.targets = tuple(targets)
        self.nobs = sum(t.lats.size for t in self.targets)
        self.mapping = mapping
        self.outmode = "stacked_array"
        self.varnames = []

    def __getstate__(self):
        self.engine.close_cashed_stores()
        return self.__dict__

    def __setstate__(self, state):
        self.__dict__.update(state)

    def make_node(self, inputs):
        """records the variable order like the reference's make_node (pytensorf.py:64-86); the
        symbolic graph node itself needs pytensor"""
        self.varnames = list(inputs.keys())
        return self

    def perform(self, node, inputs, output):
        from . import heart
        _update_sources(self.sources, self.mapping, self.varnames, inputs)
        output[0][0] = heart.geo_synthetics(engine=self.engine, targets=self.targets,
                                            sources=self.sources, outmode=self.outmode)

    def __call__(self, inputs):
        self.make_node(inputs)
        out = [[None]]
        self.perform(None, list(inputs.values()), out)
        return out[0][0]

    def infer_shape(self, fgraph=None, node=None, input_shapes=None):
        return [(self.nobs, 3)]


class SeisSynthesizer(object):
    __props__ = ("engine", "sources", "mapping", "targets", "events", "event_idx", "arrival_taper",
                 "arrival_times", "wavename", "filterer", "pre_stack_cut", "station_corrections",
                 "domain")

    def __init__(self, engine, sources, mapping, targets, events, event_idx, arrival_taper,
                 arrival_times, wavename, filterer, pre_stack_cut, station_corrections, domain):
        self.engine, self.sources, self.mapping = engine, tuple(sources), mapping
        self.targets, self.events, self.event_idx = tuple(targets), tuple(events), event_idx
        self.arrival_taper = arrival_taper
        self.arrival_times = tuple(np.asarray(arrival_times).tolist())
        self.wavename, self.filterer = wavename, tuple(filterer or ())
        self.pre_stack_cut, self.station_corrections, self.domain = pre_stack_cut, station_corrections, domain
        self.varnames = []

    def make_node(self, inputs):
        self.varnames = list(inputs.keys())
        return self

    def perform(self, node, inputs, output):
        """pytensorf.py:243-300: two outputs, synthetics (n_targets, n_samples) and their tmins"""
        from . import heart
        _update_sources(self.sources, self.mapping, self.varnames, inputs)
        output[0][0], output[1][0] = heart.seis_synthetics(
            engine=self.engine, sources=self.sources, targets=self.targets,
            arrival_taper=self.arrival_taper, wavename=self.wavename, filterer=self.filterer,
            pre_stack_cut=self.pre_stack_cut, arrival_times=np.array(self.arrival_times), outmode="array")
