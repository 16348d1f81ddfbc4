"""`layers` — the top-level package name the reference's scripts import (contextflow/model.py:14-15:
`from layers.rtdl.nn._embeddings import *`, `from layers import *`), bound to contextflow_amd.layers.

This directory holds NOTHING but this package, so putting it on sys.path exposes no other top-level name
(`model`, `build`, `dist` stay inside `contextflow_amd`).  `contextflow_amd.run` puts it ahead of the
script directory; see INTEGRATION.md section 2.

Every `layers.X` is THE SAME module object as `contextflow_amd.layers.X` (one set of classes: isinstance
checks between code that imported either name keep working); `layers.rtdl.nn._embeddings` exports the three
context encoders `create_model` takes from the reference's vendored rtdl (model.py:34,39,46)."""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

import contextflow_amd.layers as _impl
from contextflow_amd.layers import *  # noqa: F401,F403

_REAL = "contextflow_amd.layers"
_OWN = (__name__ + ".rtdl",)             # sub-packages that live in this directory


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`layers.X` -> the module `contextflow_amd.layers.X` (imported on demand, never a second copy)."""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(__name__ + ".") or fullname.startswith(_OWN):
            return None
        real = _REAL + fullname[len(__name__):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.machinery.ModuleSpec(fullname, self)

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(__name__):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules.setdefault(__name__ + _name[len(_REAL):], _mod)

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
