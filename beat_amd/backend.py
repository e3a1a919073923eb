"""
Trace files readable by the unmodified BEAT tool-chain (``beat summarize / plot``):
writers and readers for the reference's ``NumpyChain`` (".bin": one JSON header line + packed
structured records, beat/backend.py:651-898) and ``TextChain`` (".csv", :457-648), and the stage
directory naming of ``SampleStage`` (:985-1047).  Host I/O only.
"""
import itertools
import json
import os
from collections import OrderedDict

import numpy as np


def flat_names_of(varname, shape):
    """pymc ``ttab.create_flat_names``: "name__i_j" per element, the bare name for scalars"""
    if len(shape) == 0:
        return [varname]
    idx = itertools.product(*[range(s) for s in shape])
    return ["%s__%s" % (varname, "_".join(str(i) for i in ix)) for ix in idx]


class _FileChain(object):
    def __init__(self, dir_path, var_shapes=None, var_dtypes=None, buffer_size=5000, buffer_thinning=1):
        self.dir_path = dir_path
        self.var_shapes = OrderedDict((k, tuple(v)) for k, v in (var_shapes or {}).items())
        self.var_dtypes = OrderedDict((k, str((var_dtypes or {}).get(k, "float64")))
                                      for k in self.var_shapes)
        self.varnames = list(self.var_shapes.keys())
        self.flat_names = OrderedDict((k, flat_names_of(k, s)) for k, s in self.var_shapes.items())
        self.buffer, self.buffer_size, self.buffer_thinning = [], buffer_size, buffer_thinning
        self.count, self.chain, self.filename, self._df = 0, None, None, None
        os.makedirs(dir_path, exist_ok=True)

    def buffer_write(self, lpoint, draw):
        """backend.py:365-384"""
        self.count += 1
        self.buffer.append((lpoint, draw))
        if len(self.buffer) >= self.buffer_size:
            self.record_buffer()

    def record_buffer(self):
        buf = self.buffer[::self.buffer_thinning] if self.buffer_thinning > 1 else self.buffer
        if self.buffer_thinning > 1 and self.buffer and buf[-1] is not self.buffer[-1]:
            buf = buf + [self.buffer[-1]]  # thin_buffer(..., ensure_last=True)
        for lpoint, _ in buf:
            self._write(lpoint)
        self.buffer = []
        self._df = None

    def write(self, lpoint, draw=0):
        self._write(lpoint)
        self.count += 1
        self._df = None

    def __len__(self):
        return len(self._load()) if self.filename and os.path.exists(self.filename) else 0


class NumpyChain(_FileChain):
    """backend.py:651-898"""

    flat_names_tag, var_shape_tag, var_dtypes_tag = "flat_names", "var_shapes", "var_dtypes"

    @property
    def data_structure(self):
        formats = ["{shape}{dtype}".format(shape=self.var_shapes[n], dtype=self.var_dtypes[n])
                   for n in self.varnames]
        return np.dtype({"names": self.varnames, "formats": formats})

    def setup(self, draws, chain, overwrite=False):
        self.chain, self.draws = chain, draws
        self.filename = os.path.join(self.dir_path, "chain-{}.bin".format(chain))
        if os.path.exists(self.filename) and not overwrite:
            return
        self.count = 0
        header = OrderedDict([
            (self.flat_names_tag, self.flat_names),
            (self.var_shape_tag, OrderedDict((k, list(v)) for k, v in self.var_shapes.items())),
            (self.var_dtypes_tag, self.var_dtypes),
        ])
        with open(self.filename, "wb") as fh:
            fh.write((json.dumps(header) + "\n").encode())

    def _write(self, lpoint):
        data = np.zeros(1, dtype=self.data_structure)
        for name, array in zip(self.varnames, lpoint):
            data[name] = array
        with open(self.filename, mode="ab+") as fh:
            data.tofile(fh)

    def write_block(self, columns):
        """All draws of a chain at once: columns = dict name -> (ndraws, *shape)"""
        n = len(next(iter(columns.values())))
        data = np.zeros(n, dtype=self.data_structure)
        for name in self.varnames:
            data[name] = np.asarray(columns[name]).reshape((n,) + self.var_shapes[name])
        with open(self.filename, mode="ab+") as fh:
            data.tofile(fh)
        self.count += n
        self._df = None

    @classmethod
    def load(cls, filename):
        with open(filename, "rb") as fh:
            hdr = json.loads(fh.readline().decode(), object_pairs_hook=OrderedDict)
        ch = cls(os.path.dirname(filename) or ".",
                 OrderedDict((k, tuple(v)) for k, v in hdr[cls.var_shape_tag].items()),
                 hdr[cls.var_dtypes_tag])
        ch.filename = filename
        return ch

    def _load(self):
        if self._df is None:
            with open(self.filename, "rb") as fh:
                next(fh)
                self._df = np.fromfile(fh, dtype=self.data_structure)
        return self._df

    def get_values(self, varname, burn=0, thin=1):
        df = self._load()
        if varname not in self.varnames:
            raise ValueError('Did not find varname "%s" in sampling results! Fixed?' % varname)
        shape = (df.shape[0],) + self.var_shapes[varname]
        return df[varname].ravel().reshape(shape)[burn::thin]

    def point(self, idx):
        df = self._load()
        return {v: df[v][int(idx)].reshape(self.var_shapes[v]) for v in self.varnames}


class TextChain(_FileChain):
    """backend.py:457-648"""

    def setup(self, draws, chain, overwrite=False):
        self.chain, self.draws = chain, draws
        self.filename = os.path.join(self.dir_path, "chain-{}.csv".format(chain))
        if os.path.exists(self.filename) and not overwrite:
            return
        self.count = 0
        cnames = [fv for v in self.varnames for fv in self.flat_names[v]]
        with open(self.filename, "w") as fh:
            fh.write(",".join(cnames) + "\n")

    def _write(self, lpoint):
        columns = itertools.chain.from_iterable(map(str, np.asarray(v).ravel()) for v in lpoint)
        with open(self.filename, mode="a+") as fh:
            fh.write(",".join(columns) + "\n")

    def _load(self):
        if self._df is None:
            self._df = np.atleast_2d(np.loadtxt(self.filename, delimiter=",", skiprows=1))
        return self._df

    def get_values(self, varname, burn=0, thin=1):
        df = self._load()
        o = 0
        for v in self.varnames:
            n = len(self.flat_names[v])
            if v == varname:
                return df[:, o:o + n].reshape((df.shape[0],) + self.var_shapes[v])[burn::thin]
            o += n
        raise ValueError('Did not find varname "%s" in sampling results! Fixed?' % varname)


backend_catalog = {"csv": TextChain, "bin": NumpyChain}


def stage_path(homepath, stage):
    """backend.py:1003-1012 SampleStage.stage_path: stage_<k>, stage_final for -1"""
    return os.path.join(homepath, "stage_final" if stage == -1 else "stage_{}".format(stage))


def population_shapes(layout, out_names):
    """variables of a stage's traces in file order -- the free variables in layout order, then the likelihood block
    grouped into the reference's deterministics (seis_like.., geo_like.., laplacian_like, like) -> (shapes, groups)"""
    shapes = OrderedDict((k, (n,)) for k, n in layout.varsizes.items())
    groups = OrderedDict()
    for i, name in enumerate(out_names):
        key = ("seis_like" if name.startswith("seis_like")
               else "geo_like" if name.startswith("geo_like") else name)
        groups.setdefault(key, []).append(i)
    for k, idx in groups.items():
        shapes[k] = () if k in ("like", "laplacian_like") else (len(idx),)
    return shapes, groups


def thinned_draws(n_steps, thinning):
    """indices (0-based, within a stage) of the draws the reference's trace buffer keeps: every ``thinning``-th from the
    first on, and the last one (beat/backend.py:365-404 record_buffer -> utility.thin_buffer(buffer, thinning,
    ensure_last=True); the buffer holds the whole stage when buffer_size >= n_steps)"""
    n_steps, thinning = int(n_steps), max(1, int(thinning))
    keep = list(range(0, n_steps, thinning))
    if keep and keep[-1] != n_steps - 1:
        keep.append(n_steps - 1)
    return keep


def write_population(homepath, stage, layout, out_names, population, lpoints, backend="bin", n_threads=8, first_chain=0):
    """The traces of the chains of a stage: variables in layout order, then the likelihood block (seis_like..,
    geo_like.., laplacian_like, like) -- what the reference's workers leave on disk (sampler/base.py:310-311,
    316-395) and ``select_end_points`` reads back.  population (chains, nparams) / lpoints (chains, nllk): the END
    POINTS as one-draw traces; (draws, chains, nparams) / (draws, chains, nllk): every kept draw of every chain, in
    order (round 6: ``smc_sample(buffer_thinning=...)``).  ``first_chain``: global index of the first chain (a rank
    writes the files of ITS block of chains).

    "bin" (NumpyChain, beat/backend.py:651-898): the records of ALL chains are packed in one structured array (no
    per-draw Python), the JSON header line is built once, and every chain file is written by one ``open`` / one
    ``write`` from a small thread pool -- byte for byte what ``NumpyChain.setup`` + ``.write`` leave per chain
    (tests/test_backend.py), ~20 x faster at population scale (4096 chains: seconds -> tenths of a second)."""
    shapes, groups = population_shapes(layout, out_names)
    path = stage_path(homepath, stage)
    population, lpoints = np.asarray(population), np.asarray(lpoints)
    if population.ndim == 2:
        population, lpoints = population[None], lpoints[None]
    n_draws, n_chains = population.shape[0], population.shape[1]
    first_chain = int(first_chain)
    if backend != "bin":
        for c in range(n_chains):
            ch = backend_catalog[backend](path, shapes)
            ch.setup(n_draws, first_chain + c, overwrite=True)
            for j in range(n_draws):
                pt = layout.rmap(population[j, c])
                lp = [pt[k] for k in layout.varsizes]
                lp += [lpoints[j, c, idx] if k in ("seis_like", "geo_like") else lpoints[j, c, idx[0]]
                       for k, idx in groups.items()]
                ch.write(lp)
        return path
    proto = NumpyChain(path, shapes)
    header = (json.dumps(OrderedDict([
        (proto.flat_names_tag, proto.flat_names),
        (proto.var_shape_tag, OrderedDict((k, list(v)) for k, v in proto.var_shapes.items())),
        (proto.var_dtypes_tag, proto.var_dtypes)])) + "\n").encode()
    os.makedirs(path, exist_ok=True)
    data = np.zeros((n_chains, n_draws), dtype=proto.data_structure)      # a chain's records are consecutive
    for k in layout.varsizes:
        o = layout.offset(k)
        data[k] = population[:, :, o:o + layout.varsizes[k]].transpose(1, 0, 2)
    for k, idx in groups.items():
        data[k] = lpoints[:, :, idx].transpose(1, 0, 2) if k in ("seis_like", "geo_like") else lpoints[:, :, idx[0]].T
    rec = data.view(np.uint8).reshape(n_chains, n_draws * data.dtype.itemsize)

    def write_some(first):
        for c in range(first, n_chains, n_threads):
            with open(os.path.join(path, "chain-{}.bin".format(first_chain + c)), "wb") as fh:
                fh.write(header + rec[c].tobytes())
    if n_threads <= 1 or n_chains < 64:
        n_threads = 1
        write_some(0)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(n_threads) as pool:
            list(pool.map(write_some, range(n_threads)))
    return path
