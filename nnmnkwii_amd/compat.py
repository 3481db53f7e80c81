"""Route an existing ``nnmnkwii`` code base through the MI355X kernels without touching its call sites.

    import nnmnkwii_amd.compat as compat
    compat.install()
    from nnmnkwii.paramgen import mlpg            # now the HIP path
    from nnmnkwii.preprocessing.alignment import DTWAligner

Two situations:

* the reference package is importable: the hot-path callables are replaced IN PLACE on its modules
  (``nnmnkwii.paramgen.mlpg`` and friends, the autograd functions, the aligners, ``baseline.gmm.MLPG``,
  ``delta_features`` / ``trim_zeros_frames`` / the modspec functions, ``util.apply_each2d_*``);
  everything else of the reference (datasets, frontend, IO ...) keeps working as before;
* it is not: lightweight ``nnmnkwii`` / ``nnmnkwii.<sub>`` module objects are registered in
  ``sys.modules`` that expose exactly the hot-path surface of SURVEY.md 8(b) and 8(f).

``uninstall()`` restores what ``install()`` replaced.
"""
import importlib
import sys
import types

from . import autograd as _autograd
from . import paramgen as _paramgen
from . import preprocessing as _preprocessing
from . import util as _util
from .baseline import gmm as _gmm
from .preprocessing import alignment as _alignment

_modspec = importlib.import_module(__package__ + ".preprocessing.modspec")   # the attribute of that name is the function

# reference module -> {attribute: replacement}
_SURFACE = {
    "nnmnkwii.paramgen": {k: getattr(_paramgen, k) for k in (
        "mlpg", "mlpg_grad", "unit_variance_mlpg_matrix", "reshape_means", "build_win_mats", "full_window_mat")},
    "nnmnkwii.autograd": {k: getattr(_autograd, k) for k in (
        "mlpg", "MLPG", "unit_variance_mlpg", "UnitVarianceMLPG", "modspec", "ModSpec")},
    "nnmnkwii.preprocessing": {k: getattr(_preprocessing, k) for k in (
        "delta_features", "trim_zeros_frames", "modspec", "modphase", "inv_modspec", "modspec_smoothing")},
    "nnmnkwii.preprocessing.alignment": {k: getattr(_alignment, k) for k in ("DTWAligner", "IterativeDTWAligner")},
    "nnmnkwii.preprocessing.modspec": {k: getattr(_modspec, k) for k in (
        "modspec", "modphase", "inv_modspec", "modspec_smoothing")},
    "nnmnkwii.baseline.gmm": {k: getattr(_gmm, k) for k in ("MLPGBase", "MLPG")},
    "nnmnkwii.util": {"apply_each2d_padded": _util.apply_each2d_padded, "apply_each2d_trim": _util.apply_each2d_trim,
                      "trim_zeros_frames": _preprocessing.trim_zeros_frames,
                      "delta_features": _preprocessing.delta_features,
                      "apply_delta_windows": _preprocessing.delta_features},
}
_saved = []        # (module, attribute, previous value or _MISSING)
_registered = []   # names we put into sys.modules
_MISSING = object()


def _reference_available():
    try:
        importlib.import_module("nnmnkwii")
        return True
    except Exception:  # not installed, or its own dependencies (fastdtw, pysptk ...) are missing
        return False


def install():
    """Patch (or provide) the ``nnmnkwii`` hot-path surface. Idempotent."""
    if _saved or _registered:
        return
    have_ref = _reference_available()
    for modname, attrs in _SURFACE.items():
        mod = None
        if have_ref:
            try:
                mod = importlib.import_module(modname)
            except Exception:   # e.g. nnmnkwii.preprocessing.alignment without fastdtw installed
                mod = None
        if mod is None:
            mod = sys.modules.get(modname)
            if mod is None:
                mod = types.ModuleType(modname)
                mod.__doc__ = "nnmnkwii_amd.compat stand-in for %s (hot-path surface only)" % modname
                if modname.count(".") < 2 and modname != "nnmnkwii.util":
                    mod.__path__ = []   # a package: submodules may be registered under it
                sys.modules[modname] = mod
                _registered.append(modname)
                parent, _, child = modname.rpartition(".")
                if parent not in sys.modules:
                    pm = types.ModuleType(parent)
                    pm.__path__ = []
                    sys.modules[parent] = pm
                    _registered.append(parent)
                setattr(sys.modules[parent], child, mod)
        for name, obj in attrs.items():
            _saved.append((mod, name, getattr(mod, name, _MISSING)))
            setattr(mod, name, obj)


def uninstall():
    """Undo :func:`install`."""
    while _saved:
        mod, name, prev = _saved.pop()
        if prev is _MISSING:
            try:
                delattr(mod, name)
            except AttributeError:
                pass
        else:
            setattr(mod, name, prev)
    while _registered:
        sys.modules.pop(_registered.pop(), None)
