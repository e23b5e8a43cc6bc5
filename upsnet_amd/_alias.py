"""Import-path aliasing for the reference's package names.

The reference's callers import the hot path by MODULE PATH -- `upsnet.operators.modules.deform_conv`, `upsnet.nms.nms`,
`upsnet.models`, `lib.utils.data_parallel`, ... (upsnet/models/resnet_upsnet.py:22-32, rcnn.py:23-26, fcn.py:22-23,
operators/functions/pyramid_proposal.py:18-20, upsnet_end2end_test.py:31-47). The repo-root packages `upsnet/` and `lib/` install one
finder each that resolves `upsnet.X` to the ALREADY-IMPORTED-OR-IMPORTABLE module `upsnet_amd.X` (`lib.utils.X` to
`upsnet_amd.utils.X`) and registers the SAME module object under both names: one `config` singleton, one set of classes
(isinstance works across both spellings), no re-export files to keep in sync.
"""
import importlib
import importlib.abc
import importlib.util
import sys


class AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, alias, real):
        self.alias, self.real = alias, real

    def _real_name(self, fullname):
        return self.real + fullname[len(self.alias):]

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(self.alias + '.'):
            return None
        try:
            spec = importlib.util.find_spec(self._real_name(fullname))
        except (ImportError, ValueError):
            return None
        if spec is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=spec.submodule_search_locations is not None)

    def create_module(self, spec):
        real = importlib.import_module(self._real_name(spec.name))   # the same module object under the second name
        # importlib is about to overwrite __spec__ / __loader__ / __package__ / __path__ of the object this returns with the ALIAS spec's
        # values (_init_module_attrs) -- on the real module that would break importlib.reload() (exec_module below does nothing)
        # and relative imports inside upsnet_amd (`__package__ != __spec__.parent`). Remember the real ones; exec_module puts them back.
        self._saved = getattr(self, '_saved', {})
        self._saved[spec.name] = {k: getattr(real, k) for k in ('__spec__', '__loader__', '__package__', '__path__', '__file__', '__cached__')
                                  if hasattr(real, k)}
        return real

    def exec_module(self, module):
        saved = getattr(self, '_saved', {})
        for name in [n for n in saved if sys.modules.get(self._real_name(n)) is module]:
            for k, v in saved.pop(name).items():
                setattr(module, k, v)


def install(alias, real):
    if not any(isinstance(f, AliasFinder) and f.alias == alias for f in sys.meta_path):
        sys.meta_path.insert(0, AliasFinder(alias, real))
