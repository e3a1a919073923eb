"""
GF libraries for the distributed-slip (FFI) forward model -- HBM-resident counterparts of
``beat.ffi.base`` (reference beat/ffi/base.py).

Same construction and call signatures as the reference classes:

  SeismicGFLibrary(config).setup(ntargets, npatches, ndurations, nstarttimes, nsamples,
                                 allocate=True); .put(...); .init_optimization();
      .stack_all(durations, starttimes, slips, targetidxs, patchidxs, interpolation)
                                                                  base.py:320-709
  GeodeticGFLibrary(config).setup(npatches, nsamples, allocate=True); .put(...);
      .stack_all(slips)                                           base.py:192-317

Differences by design: the stacking runs on the GPU (stack mode "hip"; there is no CPU
fallback), the library lives once in HBM instead of in a fork-shared RawArray
(parallel.memshare), and ``stack_all_batch`` evaluates many chains per call.
"""
import os

import numpy as np

from .engine import get_context

gf_dtype = "float64"  # base.py:18


class GFLibraryError(Exception):
    pass


class SeismicGFLibraryConfig(object):
    """The fields of beat/config.py:1900-1926 the stacking path needs."""

    def __init__(self, dimensions=(0, 0, 0, 0, 0), starttime_sampling=0.5, duration_sampling=0.5,
                 starttime_min=0.0, duration_min=0.1, component="uparr", datatype="seismic",
                 mapnumber=1, wavename="any_P", crust_ind=0):
        self.dimensions = tuple(int(d) for d in dimensions)
        self.starttime_sampling = float(starttime_sampling)
        self.duration_sampling = float(duration_sampling)
        self.starttime_min = float(starttime_min)
        self.duration_min = float(duration_min)
        self.component = component
        self.datatype = datatype
        self.mapnumber = mapnumber
        self.wavename = wavename
        self.crust_ind = crust_ind

    @property
    def _mapid(self):
        return "_".join((self.wavename, str(self.mapnumber)))


class GeodeticGFLibraryConfig(object):
    def __init__(self, dimensions=(0, 0), component="uparr", datatype="geodetic", crust_ind=0):
        self.dimensions = tuple(int(d) for d in dimensions)
        self.component = component
        self.datatype = datatype
        self.crust_ind = crust_ind


def get_gf_prefix(datatype, component, wavename, crust_ind):
    """base.py:155-156"""
    return "%s_%s_%s_%i" % (datatype, component, wavename, crust_ind)


class GFLibrary(object):
    def __init__(self, config):
        self.config = config
        self._gfmatrix = None  # host staging copy (optional)
        self._mode = "hip"
        self._ctx = None
        self.lib_id = None

    def set_stack_mode(self, mode="hip"):
        """base.py:128-147.  Only the GPU mode exists here."""
        if mode not in ("hip",):
            raise GFLibraryError(
                "Stacking mode %s not available! Available modes: hip (no CPU fallback)" % mode)
        self._mode = mode

    def get_stack_mode(self):
        return self._mode

    def _check_setup(self):
        if sum(self.config.dimensions) == 0:
            raise GFLibraryError("%s Greens Function Library is not set up!" % self.datatype)

    @property
    def size(self):
        return int(np.array(self.config.dimensions).prod())

    @property
    def filesize(self):
        return self.size * 8.0 / (1024.0 ** 2)

    @property
    def datatype(self):
        return self.config.datatype


class SeismicGFLibrary(GFLibrary):
    """5-D float64 library (ntargets, npatches, ndurations, nstarttimes, nsamples), C-order,
    sample axis fastest: one (target, patch, duration, starttime) trace = one contiguous row."""

    def __init__(self, config=None):
        super(SeismicGFLibrary, self).__init__(config or SeismicGFLibraryConfig())
        self._tmins = None
        self._device_tensor = None  # keeps an adopted torch tensor alive

    # -- reference properties (base.py:376-470)
    @property
    def dimensions(self):
        return self.config.dimensions

    @dimensions.setter
    def dimensions(self, d):
        self.config.dimensions = tuple(int(x) for x in d)

    ntargets = property(lambda self: self.config.dimensions[0])
    npatches = property(lambda self: self.config.dimensions[1])
    ndurations = property(lambda self: self.config.dimensions[2])
    nstarttimes = property(lambda self: self.config.dimensions[3])
    nsamples = property(lambda self: self.config.dimensions[4])
    starttime_min = property(lambda self: self.config.starttime_min)
    starttime_sampling = property(lambda self: self.config.starttime_sampling)
    duration_min = property(lambda self: self.config.duration_min)
    duration_sampling = property(lambda self: self.config.duration_sampling)

    @property
    def filename(self):
        return get_gf_prefix(self.config.datatype, self.config.component, self.config._mapid,
                             self.config.crust_ind)

    @property
    def patchidxs(self):
        return np.arange(self.npatches, dtype="int16")

    def setup(self, ntargets, npatches, ndurations, nstarttimes, nsamples, allocate=False):
        """base.py:375-385"""
        self.dimensions = (ntargets, npatches, ndurations, nstarttimes, nsamples)
        if allocate:
            self._gfmatrix = np.zeros(self.dimensions)
            self._tmins = np.zeros([ntargets])

    def set_patch_time(self, targetidx, tmin):
        if self._tmins is None:
            raise GFLibraryError("Neither shared nor standard GFLibrary is setup!")
        self._tmins[targetidx] = tmin

    def put(self, entries, targetidx, patchidx, durations, starttimes):
        """base.py:422-471: fill traces of one (target, patch) for all durations x starttimes"""
        entries = np.asarray(entries)
        if len(entries.shape) < 2:
            raise ValueError("Entries have to be 2d arrays!")
        if entries.shape[1] != self.nsamples:
            raise GFLibraryError(
                "Trace length of entries is not consistent with the library"
                " to be filled! Entries length: %i Library: %i." % (entries.shape[1], self.nsamples))
        self._check_setup()
        if self._gfmatrix is None:
            raise GFLibraryError("Neither shared nor standard GFLibrary is setup!")
        durationidxs, _ = self.durations2idxs(durations)
        starttimeidxs, _ = self.starttimes2idxs(starttimes)
        self._gfmatrix[targetidx, patchidx, durationidxs, starttimeidxs, :] = entries
        self.lib_id_dirty = True

    # -- index maps (host twins of the device code; base.py:486-568)
    def starttimes2idxs(self, starttimes, interpolation="nearest_neighbor"):
        return _time2idx(starttimes, self.starttime_min, self.starttime_sampling, interpolation)

    def durations2idxs(self, durations, interpolation="nearest_neighbor"):
        return _time2idx(durations, self.duration_min, self.duration_sampling, interpolation)

    def idxs2durations(self, idxs):
        return idxs * self.duration_sampling + self.duration_min

    def idxs2starttimes(self, idxs):
        return idxs * self.starttime_sampling + self.starttime_min

    # -- HBM residency
    def init_optimization(self, ctx=None):
        """base.py:387-404: make the library available to the sampler -> upload to HBM once."""
        ctx = ctx or get_context()
        if self.lib_id is not None and self._ctx is ctx and not getattr(self, "lib_id_dirty", False):
            return
        self._check_setup()
        if self.lib_id is not None:
            self._ctx.seis_gflib_destroy(self.lib_id)
        self._ctx = ctx
        self.lib_id = ctx.seis_gflib_create(self.dimensions, self.starttime_min,
                                            self.starttime_sampling, self.duration_min,
                                            self.duration_sampling)
        if self._device_tensor is not None:
            ctx.seis_gflib_adopt(self.lib_id, self._device_tensor)
        else:
            if self._gfmatrix is None:
                raise GFLibraryError("Neither shared nor standard GFLibrary is setup!")
            # chunked so that a 60 GB library does not need a second host copy
            flat = self._gfmatrix.reshape(-1)
            step = 1 << 27  # 1 GiB of doubles
            for o in range(0, flat.size, step):
                ctx.seis_gflib_upload(self.lib_id, flat[o:o + step], o)
        self.lib_id_dirty = False

    def adopt_device_tensor(self, tensor):
        """Use an existing torch CUDA float64 tensor of shape ``dimensions`` as the library
        (no host copy; e.g. libraries generated or loaded directly into HBM)."""
        if tuple(tensor.shape) != tuple(self.dimensions):
            raise GFLibraryError("tensor shape %s != library dimensions %s"
                                 % (tuple(tensor.shape), self.dimensions))
        self._device_tensor = tensor.contiguous()
        self.lib_id_dirty = True

    def get_all(self):
        return self._gfmatrix

    def save(self, outdir="", filename=None):
        """base.py:364-373 on-disk format: <name>.traces.npy / <name>.times.npy"""
        filename = filename or self.filename
        outpath = os.path.join(outdir, filename)
        np.save(outpath + ".traces", arr=self._gfmatrix, allow_pickle=False)
        np.save(outpath + ".times", arr=self._tmins, allow_pickle=False)

    def load(self, outdir, filename=None):
        """base.py:161-189 load_gf_library (array part)"""
        filename = filename or self.filename
        outpath = os.path.join(outdir, filename)
        self._gfmatrix = np.load(outpath + ".traces.npy", allow_pickle=False)
        self._tmins = np.load(outpath + ".times.npy", allow_pickle=False)
        self.dimensions = self._gfmatrix.shape
        self.lib_id_dirty = True

    # -- stacking
    def stack_all(self, durations, starttimes, slips, targetidxs=None, patchidxs=None,
                  interpolation="nearest_neighbor"):
        """base.py:607-709, one chain.  -> (ntargets, nsamples)"""
        if targetidxs is None:
            raise ValueError("Target indexes have to be defined!")
        targetidxs = np.asarray(targetidxs).ravel()
        if patchidxs is not None and not np.array_equal(np.asarray(patchidxs).ravel(), self.patchidxs):
            raise NotImplementedError("stacking a subset of patches is not supported on the GPU path")
        T, P = self.ntargets, self.npatches
        st = np.broadcast_to(np.asarray(starttimes, dtype=np.float64), (T, P))
        out = self.stack_all_batch(np.asarray(durations, dtype=np.float64).reshape(1, P),
                                   st.reshape(1, T, P),
                                   np.asarray(slips, dtype=np.float64).reshape(1, P),
                                   interpolation=interpolation)[0]
        if not np.array_equal(targetidxs, np.arange(T)):
            out = out[targetidxs]
        return out

    def stack_all_batch(self, durations, starttimes, slips, interpolation="nearest_neighbor"):
        """durations (C,P), starttimes (C,T,P), slips (C,P) -> (C,T,N); numpy or torch-cuda."""
        self.init_optimization(self._ctx)
        return self._ctx.seis_stack_all_batch(self.lib_id, self.dimensions, durations, starttimes,
                                              slips, interpolation)

    def stack(self, targetidx, patchidxs, durations, starttimes, slips,
              interpolation="nearest_neighbor"):
        """base.py:570-605 single target"""
        T, P = self.ntargets, self.npatches
        st = np.zeros((T, P))
        st[targetidx] = starttimes
        return self.stack_all(durations, st, slips, targetidxs=np.arange(T),
                              interpolation=interpolation)[targetidx]


def _time2idx(x, xmin, dx, interpolation):
    """base.py:486-568"""
    x = np.asarray(x, dtype=np.float64)
    if interpolation == "nearest_neighbor":
        return np.round((x - xmin) / dx).astype("int16"), None
    elif interpolation == "multilinear":
        d = (x - xmin) / dx
        c = np.ceil(d).astype("int16")
        return c, c - d
    raise NotImplementedError("Interpolation scheme %s not implemented!" % interpolation)


class GeodeticGFLibrary(GFLibrary):
    """(npatches, nsamples) static library; stack_all(slips) = G.T.dot(slips) (base.py:292-305)"""

    def __init__(self, config=None):
        super(GeodeticGFLibrary, self).__init__(config or GeodeticGFLibraryConfig())

    npatches = property(lambda self: self.config.dimensions[0])
    nsamples = property(lambda self: self.config.dimensions[1])

    @property
    def filename(self):
        return get_gf_prefix(self.config.datatype, self.config.component, "static",
                             self.config.crust_ind)

    def setup(self, npatches, nsamples, allocate=False):
        self.config.dimensions = (int(npatches), int(nsamples))
        if allocate:
            self._gfmatrix = np.zeros(self.config.dimensions)

    def put(self, entries, patchidx):
        """base.py:259-290"""
        if self._gfmatrix is None:
            raise GFLibraryError("Neither shared nor standard GFLibrary is setup!")
        entries = np.asarray(entries)
        if entries.shape[-1] != self.nsamples:
            raise GFLibraryError("Trace length of entries is not consistent with the library!")
        self._gfmatrix[patchidx, :] = entries
        self.lib_id_dirty = True

    def init_optimization(self, ctx=None):
        ctx = ctx or get_context()
        if self.lib_id is not None and self._ctx is ctx and not getattr(self, "lib_id_dirty", False):
            return
        if self._gfmatrix is None:
            raise GFLibraryError("Neither shared nor standard GFLibrary is setup!")
        if self.lib_id is not None:
            self._ctx.geo_gflib_destroy(self.lib_id)
        self._ctx = ctx
        self.lib_id = ctx.geo_gflib_create(self._gfmatrix)
        self.lib_id_dirty = False

    def get_all(self):
        return self._gfmatrix

    def stack_all(self, slips):
        s = np.asarray(slips, dtype=np.float64).reshape(1, -1)
        return self.stack_all_batch(s)[0]

    def stack_all_batch(self, slips, out=None):
        self.init_optimization(self._ctx)
        return self._ctx.geo_stack_all_batch(self.lib_id, self.nsamples, slips, out)
