"""Loader for the reference's own MLPG modules compiled into oracle/_ref/ -- TEST
INFRASTRUCTURE ONLY (tests, smoke(), bench.py's cpu_baseline of kind "reference").

oracle/_ref/ holds nothing but extension modules built by oracle/build_ref_so.sh from the
sources under /root/reference (Cython -> C -> .so; no source is copied).  The package objects
``nnmnkwii``, ``nnmnkwii.paramgen`` and ``nnmnkwii.util`` are synthesised here (their real
``__init__`` modules drag in sklearn, datasets, ...), so that ``nnmnkwii.paramgen._mlpg`` -- the
reference's own ``mlpg`` / ``mlpg_grad`` / ``unit_variance_mlpg_matrix`` -- imports unchanged.
"""
import importlib
import os
import sys
import types

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def available():
    return os.path.isdir(os.path.join(_ROOT, "nnmnkwii", "paramgen"))


def load():
    """Return the reference's ``nnmnkwii.paramgen._mlpg`` module (compiled), or raise ImportError."""
    if not available():
        raise ImportError("oracle/_ref is not built (run oracle/build_ref_so.sh where /root/reference exists)")
    if "nnmnkwii" in sys.modules and not getattr(sys.modules["nnmnkwii"], "_oracle_ref", False):
        raise ImportError("a different `nnmnkwii` package is already imported")
    for name, sub in (("nnmnkwii", ""), ("nnmnkwii.paramgen", "paramgen"), ("nnmnkwii.util", "util")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(_ROOT, "nnmnkwii", sub) if sub else os.path.join(_ROOT, "nnmnkwii")]
            m.__package__ = name
            m._oracle_ref = True
            sys.modules[name] = m
    sys.modules["nnmnkwii"].paramgen = sys.modules["nnmnkwii.paramgen"]
    sys.modules["nnmnkwii"].util = sys.modules["nnmnkwii.util"]
    return importlib.import_module("nnmnkwii.paramgen._mlpg")
