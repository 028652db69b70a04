"""Drop-in import surface: `models.*` / `utils.*` as the reference's train_all.py:20-31 spells them.

INTEGRATION.md puts the directory `rehrseg_amd/` itself on sys.path, so `models` and `utils`
are found as TOP-LEVEL packages there.  Their `__init__` calls `alias()` below: the package
object registered under the top-level name is replaced by the real `rehrseg_amd.<name>` package
and a meta-path finder maps every `models.X.Y` / `utils.X` import onto `rehrseg_amd.models.X.Y` /
`rehrseg_amd.utils.X`.  There is one set of module objects (one `ops` registry, one loaded
library), whichever spelling the caller uses.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

_ALIASED = set()


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".", 1)[0] in _ALIASED:
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        real = importlib.import_module("rehrseg_amd." + spec.name)  # ImportError for names the mirror lacks
        # importlib's module_from_spec overwrites __spec__ / __loader__ of whatever create_module returns with the
        # alias spec; the real module must keep its own (its lazy relative imports resolve against
        # __spec__.parent == __package__ == "rehrseg_amd.<...>").  exec_module puts them back.
        self._real[real.__name__] = (real.__spec__, real.__loader__)
        return real

    def exec_module(self, module):
        spec, loader = self._real.pop(module.__name__, (None, None))
        if spec is not None:
            module.__spec__, module.__loader__ = spec, loader

    _real = {}


_finder = _AliasFinder()


def alias(top):
    """Make the top-level package name `top` ("models" / "utils") resolve to rehrseg_amd.<top>."""
    real = importlib.import_module("rehrseg_amd." + top)
    _ALIASED.add(top)
    if _finder not in sys.meta_path:
        sys.meta_path.insert(0, _finder)
    sys.modules[top] = real
    return real
