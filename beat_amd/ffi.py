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
        """config.py:1914-1919"""
        if self.mapnumber is None:
            return self.wavename
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
        """base.py:128-147.  BEAT switches between "numpy" (export / plotting) and "pytensor"
        (sampling) arithmetic; here every mode evaluates on the GPU -- the two reference names are
        accepted so that existing call sites keep working, and there is still no CPU fallback."""
        if mode not in ("hip", "numpy", "pytensor"):
            raise GFLibraryError(
                "Stacking mode %s not available! Available modes: hip (aliases: numpy, pytensor)" % mode)
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

    def round_to_f32(self, ctx=None):
        """ROUND THE LIBRARY IN HBM TO FLOAT-REPRESENTABLE VALUES, irreversibly (values change by up to 6e-8
        relative; an adopted caller-owned tensor is overwritten too), and keep a float copy of the same values
        for the kernels that can read it (half the row traffic).  Every kernel then sees one library whichever
        copy it reads.  Never called implicitly; reload the library to get the unrounded values back.  See
        ``LogpForwFunc.round_libraries_to_f32``."""
        self.init_optimization(ctx)
        self._ctx.seis_gflib_round_to_f32(self.lib_id)
        self.rounded_to_f32 = True

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
        """base.py:364-373 on-disk format: <name>.traces.npy / <name>.times.npy / <name>.yaml.
        A library that lives only in HBM is written through a memory-mapped .npy in 1 GiB pieces
        (no full host copy)."""
        filename = filename or self.filename
        outpath = os.path.join(outdir, filename)
        if self._gfmatrix is not None:
            np.save(outpath + ".traces", arr=self._gfmatrix, allow_pickle=False)
        elif self._device_tensor is not None:
            mm = np.lib.format.open_memmap(outpath + ".traces.npy", mode="w+", dtype=np.float64,
                                           shape=tuple(self.dimensions))
            flat_h, flat_d = mm.reshape(-1), self._device_tensor.reshape(-1)
            step = 1 << 27
            for o in range(0, flat_h.size, step):
                flat_h[o:o + step] = flat_d[o:o + step].cpu().numpy()
            mm.flush()
            del mm
        else:
            raise GFLibraryError("Neither shared nor standard GFLibrary is setup!")
        tmins = self._tmins if self._tmins is not None else np.zeros([self.ntargets])
        np.save(outpath + ".times", arr=tmins, allow_pickle=False)
        self.save_config(outdir=outdir, filename=filename)

    def save_config(self, outdir="", filename=None):
        """base.py:109-116: the library's config next to the arrays, in the guts/YAML layout of
        beat.config.SeismicGFLibraryConfig (config.py:1900-1926)"""
        filename = filename or self.filename
        with open(os.path.join(outdir, filename + ".yaml"), "w") as f:
            f.write(dump_library_config(self.config))

    def load_config(self, filename):
        """base.py:118-126"""
        self.config = load_library_config(filename)

    def load(self, outdir, filename=None):
        """base.py:161-189 load_gf_library for this object: config from the .yaml (when present),
        traces and times memory-mapped -- ``init_optimization`` then streams the traces to HBM
        in 1 GiB pieces without materialising the library in host memory."""
        filename = filename or self.filename
        outpath = os.path.join(outdir, filename)
        if os.path.exists(outpath + ".yaml"):
            self.load_config(outpath + ".yaml")
        self._gfmatrix = np.load(outpath + ".traces.npy", mmap_mode="r", allow_pickle=False)
        self._tmins = np.load(outpath + ".times.npy", mmap_mode="r", allow_pickle=False)
        if self._gfmatrix.dtype != np.float64 or self._gfmatrix.ndim != 5:
            raise GFLibraryError("%s.traces.npy is not a 5-D float64 library" % outpath)
        if sum(self.config.dimensions) and tuple(self.config.dimensions) != tuple(self._gfmatrix.shape):
            raise GFLibraryError("config dimensions %s do not match the traces %s"
                                 % (self.config.dimensions, self._gfmatrix.shape))
        self.dimensions = self._gfmatrix.shape
        self._device_tensor = None
        self.lib_id_dirty = True

    # -- stacking
    def stack_all(self, durations, starttimes, slips, targetidxs=None, patchidxs=None,
                  interpolation="nearest_neighbor"):
        """base.py:607-709, one chain.  -> (len(targetidxs), nsamples).
        ``patchidxs`` may select a subset of the patches (base.py:651-656 indexes the library
        with it): durations / starttimes / slips then refer to that subset, the other patches
        contribute nothing (slip 0 on a valid grid node)."""
        if targetidxs is None:
            raise ValueError("Target indexes have to be defined!")
        targetidxs = np.asarray(targetidxs).ravel()
        T, P = self.ntargets, self.npatches
        durations = np.asarray(durations, dtype=np.float64).ravel()
        slips = np.asarray(slips, dtype=np.float64).ravel()
        starttimes = np.asarray(starttimes, dtype=np.float64)
        if patchidxs is not None and not np.array_equal(np.asarray(patchidxs).ravel(), self.patchidxs):
            pidx = np.asarray(patchidxs).ravel().astype(np.int64)
            if pidx.size != np.unique(pidx).size or (pidx < 0).any() or (pidx >= P).any():
                raise IndexError("patchidxs must be unique indexes into the %d patches" % P)
            n = pidx.size
            d_full = np.full(P, self.duration_min)
            s_full = np.zeros(P)
            st_full = np.full((T, P), self.starttime_min)
            d_full[pidx], s_full[pidx] = durations[:n], slips[:n]
            st_sub = np.broadcast_to(starttimes, (targetidxs.size, n)) if starttimes.ndim == 2 \
                and starttimes.shape[0] == targetidxs.size else np.broadcast_to(starttimes, (T, n))
            if st_sub.shape[0] == T:
                st_full[:, pidx] = st_sub
            else:
                st_full[np.ix_(targetidxs, pidx)] = st_sub
            durations, slips, starttimes = d_full, s_full, st_full
        if starttimes.ndim == 2 and starttimes.shape[0] == targetidxs.size != T:
            full = np.full((T, P), self.starttime_min)
            full[targetidxs] = starttimes
            starttimes = full
        st = np.broadcast_to(starttimes, (T, P))
        out = self.stack_all_batch(durations.reshape(1, P), st.reshape(1, T, P), slips.reshape(1, P),
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


# ---------------------------------------------------------------- library config on disk
_SEISMIC_TAG, _GEODETIC_TAG = "!beat.SeismicGFLibraryConfig", "!beat.GeodeticGFLibraryConfig"


def dump_library_config(cfg):
    """YAML text in the layout pyrocko.guts writes for beat.config.SeismicGFLibraryConfig /
    GeodeticGFLibraryConfig (config.py:1879-1926): a tagged top-level mapping; the wave name sits
    in the nested ``wave_config``.  Fields BEAT fills with defaults on load are omitted."""
    import yaml
    if isinstance(cfg, SeismicGFLibraryConfig):
        body = dict(component=cfg.component, crust_ind=int(cfg.crust_ind),
                    starttime_sampling=cfg.starttime_sampling, duration_sampling=cfg.duration_sampling,
                    starttime_min=cfg.starttime_min, duration_min=cfg.duration_min,
                    dimensions=[int(d) for d in cfg.dimensions], datatype=cfg.datatype,
                    mapnumber=cfg.mapnumber)
        text = yaml.safe_dump(body, default_flow_style=False, sort_keys=False)
        text += "wave_config: !beat.WaveformFitConfig\n  name: %s\n" % cfg.wavename
        return "--- %s\n%s" % (_SEISMIC_TAG, text)
    body = dict(component=cfg.component, crust_ind=int(cfg.crust_ind),
                dimensions=[int(d) for d in cfg.dimensions], datatype=cfg.datatype)
    return "--- %s\n%s" % (_GEODETIC_TAG, yaml.safe_dump(body, default_flow_style=False, sort_keys=False))


def load_library_config(path):
    """Read a library .yaml written by BEAT (pyrocko.guts dump) or by ``dump_library_config``:
    application tags (!beat.*, !pf.*) are read as plain mappings; only the fields of the stacking
    path are kept (base.py:118-126, config.py:1900-1926)."""
    import yaml

    class _Loader(yaml.SafeLoader):
        pass

    def _any(loader, suffix, node):
        if isinstance(node, yaml.MappingNode):
            return loader.construct_mapping(node, deep=True)
        if isinstance(node, yaml.SequenceNode):
            return loader.construct_sequence(node, deep=True)
        return loader.construct_scalar(node)

    _Loader.add_multi_constructor("!", _any)
    with open(path) as f:
        d = yaml.load(f, Loader=_Loader)
    if not isinstance(d, dict) or "dimensions" not in d:
        raise GFLibraryError("%s is not a GF library config" % path)
    if d.get("datatype", "seismic") == "geodetic" or len(d["dimensions"]) == 2:
        return GeodeticGFLibraryConfig(dimensions=d["dimensions"], component=d.get("component", "uparr"),
                                       datatype="geodetic", crust_ind=d.get("crust_ind", 0))
    wc = d.get("wave_config") or {}
    return SeismicGFLibraryConfig(
        dimensions=d["dimensions"], starttime_sampling=d.get("starttime_sampling", 0.5),
        duration_sampling=d.get("duration_sampling", 0.5), starttime_min=d.get("starttime_min", 0.0),
        duration_min=d.get("duration_min", 0.1), component=d.get("component", "uparr"),
        datatype=d.get("datatype", "seismic"), mapnumber=d.get("mapnumber", None),
        wavename=wc.get("name", "any_P") if isinstance(wc, dict) else "any_P",
        crust_ind=d.get("crust_ind", 0))


def load_gf_library(directory="", filename=None):
    """base.py:161-189: config from <filename>.yaml, traces (and times) memory-mapped; the
    datatype is the first token of the file name"""
    inpath = os.path.join(directory, filename)
    datatype = filename.split("_")[0]
    if datatype == "seismic":
        gfs = SeismicGFLibrary()
        gfs.load_config(inpath + ".yaml")
        gfs.load(directory, filename)
    elif datatype == "geodetic":
        gfs = GeodeticGFLibrary()
        gfs.config = load_library_config(inpath + ".yaml")
        gfs._gfmatrix = np.load(inpath + ".traces.npy", mmap_mode="r", allow_pickle=False)
        gfs.lib_id_dirty = True
    else:
        raise ValueError('datatype "%s" not supported!' % datatype)
    return gfs


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

    def save(self, outdir="", filename=None):
        """base.py:246-257: <name>.traces.npy + <name>.yaml"""
        filename = filename or self.filename
        np.save(os.path.join(outdir, filename) + ".traces", arr=self._gfmatrix, allow_pickle=False)
        with open(os.path.join(outdir, filename + ".yaml"), "w") as f:
            f.write(dump_library_config(self.config))

    def stack_all(self, slips):
        s = np.asarray(slips, dtype=np.float64).reshape(1, -1)
        return self.stack_all_batch(s)[0]

    def stack_all_batch(self, slips, out=None):
        self.init_optimization(self._ctx)
        return self._ctx.geo_stack_all_batch(self.lib_id, self.nsamples, slips, out)
