"""
TEST INFRASTRUCTURE ONLY -- never imported by the product (beat_amd/).

Import harness that makes the *numpy-mode* hot path of the reference
(hvasbath/beat, mounted read-only at /root/reference) importable in this
container, where pytensor / pymc / pyrocko are not installed.

It is used by ``oracle/gen_golden.py`` to generate the golden vectors under
``tests/golden/`` from the reference's own code.  The
reference cannot travel to the GPU box, so nothing here is used at test time:
tests read the committed ``.npz`` fixtures.

How it works: a meta-path finder fabricates permissive placeholder modules for
every missing third-party package.  ``pyrocko.guts.Object`` is given just enough
behaviour (kwargs -> attributes, class-level ``X.T(default=...)`` descriptors)
for the reference's config/dataclass style objects to be constructed.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("BEAT_REFERENCE_ROOT", "/root/reference")

_STUB_ROOTS = ("pytensor", "pyrocko", "pymc", "arviz", "mpi4py", "cutde", "pygmsh")


# --------------------------------------------------------------------------- guts
class _TSpec(object):
    """What ``Float.T(default=...)`` returns: remembers the default only."""

    def __init__(self, *args, **kwargs):
        self.default = kwargs.get("default", None)
        self.optional = kwargs.get("optional", False)

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):  # permissive
        return _Anything()


class _GutsMeta(type):
    def __new__(mcs, name, bases, ns):
        specs = {}
        for b in bases:
            specs.update(getattr(b, "_t_specs", {}))
        for k, v in list(ns.items()):
            if isinstance(v, _TSpec):
                specs[k] = v
                del ns[k]
        cls = super().__new__(mcs, name, bases, ns)
        cls._t_specs = specs
        return cls

    @property
    def T(cls):
        return _TFactory(cls)


class _TFactory(object):
    def __init__(self, cls):
        self.cls = cls

    def __call__(self, *a, **k):
        return _TSpec(*a, **k)

    def __getattr__(self, name):
        return _Anything()


class GutsObject(metaclass=_GutsMeta):
    def __init__(self, **kwargs):
        import copy

        for k, spec in type(self)._t_specs.items():
            if k in kwargs:
                setattr(self, k, kwargs.pop(k))
            else:
                d = spec.default
                if callable(d) and not isinstance(d, type):
                    try:
                        d = d()
                    except Exception:
                        pass
                setattr(self, k, copy.copy(d))
        for k, v in kwargs.items():
            setattr(self, k, v)

    @classmethod
    def D(cls, **kwargs):
        return cls(**kwargs)

    def regularize(self):
        pass

    def validate(self):
        pass


class _Anything(object):
    """Callable / iterable / indexable / subclassable placeholder."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())

    def __getitem__(self, k):
        return [] if isinstance(k, slice) else _Anything()

    def __mro_entries__(self, bases):
        return (object,)

    def __bool__(self):
        return False


# ------------------------------------------------------------------- pytensor bits
class _Shared(object):
    def __init__(self, value, name=None, borrow=False, **k):
        self._v = value
        self.name = name

    def get_value(self, borrow=False):
        return self._v

    def set_value(self, v, borrow=False):
        self._v = v

    def astype(self, dtype):
        return self


def _shared(value, name=None, borrow=False, **k):
    return _Shared(value, name=name)


_NOT_SUBMODULES = {"shared", "config", "load", "dump"}


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        # class-like names become guts Objects so they can be subclassed with
        # ``X.T(...)`` descriptors; everything else is a permissive placeholder.
        if name[:1].isupper() and self.__name__.split(".")[0] == "pyrocko":
            cls = _GutsMeta(name, (GutsObject,), {})
            setattr(self, name, cls)
            return cls
        if name[:1].islower() and name not in _NOT_SUBMODULES:
            # ``from pyrocko import gf`` must yield a (stub) submodule so that
            # ``gf.Target`` is subclassable
            return importlib.import_module(full)
        val = _Anything()
        setattr(self, name, val)
        return val

    def __call__(self, *a, **k):
        return _Anything()

    def __iter__(self):
        return iter(())

    def __getitem__(self, k):
        return [] if isinstance(k, slice) else _Anything()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        name = module.__name__
        if name == "pytensor":
            module.config = types.SimpleNamespace(floatX="float64")
            module.shared = _shared
        elif name == "pytensor.tensor":
            module.Op = object
        elif name == "pytensor.graph":
            module.Apply = _Anything
        elif name == "pyrocko.guts":
            module.Object = GutsObject
            for n in (
                "Float Int String List Dict Tuple Bool StringChoice StringUnion "
                "Any Timestamp Choice Union Unicode Complex"
            ).split():
                setattr(module, n, _GutsMeta(n, (GutsObject,), {}))
            module.load = _Anything()
            module.dump = _Anything()
            module.ArgumentError = type("ArgumentError", (Exception,), {})
            module.ValidationError = type("ValidationError", (Exception,), {})
        elif name == "pyrocko.guts_array":
            module.Array = _GutsMeta("Array", (GutsObject,), {})
        elif name == "pyrocko.gf.seismosizer":
            module.Cloneable = type("Cloneable", (object,), {})
        elif name == "pymc.vartypes":
            module.discrete_types = set()
        # wire as attribute of parent
        if "." in name:
            parent, child = name.rsplit(".", 1)
            if parent in sys.modules:
                setattr(sys.modules[parent], child, module)


_installed = False


def install(fast_sweep_ext_dir=None):
    """Install the stubs and put /root/reference on sys.path (idempotent)."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.meta_path.insert(0, _StubFinder())

    info = types.ModuleType("beat.info")
    info.version = "2.0.5"
    info.project_root = REFERENCE_ROOT
    sys.modules["beat.info"] = info

    defaults = types.ModuleType("beat.defaults")
    defaults.defaults = _Anything()
    defaults.default_decimation_factors = {}
    defaults.hypername = lambda *a, **k: "h_x"
    sys.modules["beat.defaults"] = defaults

    if fast_sweep_ext_dir is None:
        fast_sweep_ext_dir = os.path.join(os.path.dirname(__file__), "_ref")
    if fast_sweep_ext_dir not in sys.path:
        sys.path.insert(0, fast_sweep_ext_dir)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def stub_unpickle(path):
    """Load a reference pickle (e.g. data/examples/Laquila/geodetic_data.pkl)
    without pyrocko: every beat/pyrocko/pytensor global becomes a bare state holder."""
    import pickle

    class Holder(object):
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, state):
            if isinstance(state, dict):
                self.__dict__.update(state)
            else:
                self.__dict__["_state"] = state

    class U(pickle.Unpickler):
        def find_class(self, module, name):
            root = module.split(".")[0]
            if root in ("beat", "pyrocko", "pytensor", "theano"):
                return type(name, (Holder,), {})
            if module.startswith("numpy.core"):
                module = module.replace("numpy.core", "numpy._core")
            return super().find_class(module, name)

    with open(path, "rb") as f:
        return U(f).load()
