"""Pickles that cross between this package and the reference: `samples.pt`, `trajectories.pt`, the starting-configuration pickle.

All of them hold `AXL` named tuples, and a pickle names a class by its module path: a file the reference wrote names
`diffusion_for_multi_scale_molecular_dynamics.namespace.AXL` (src/namespace.py:15-44), a file this package wrote names this
package's.  A process that has only one of the two packages cannot `torch.load` the other's file.

  load(path)                 reads either: classes of the reference package are looked up at the same relative path in this
                             package (utils/lightning_checkpoint.TolerantUnpickler), so a starting-configuration pickle made with
                             the reference's tools (generators/trajectory_initializer.py:151-161) or its `samples.pt` load here.
  save_for_reference(obj, path)   writes `obj` with every named tuple of this package (AXL; the recorder's `Noise` tables) named as the
                             class of the same name at the same relative path of the REFERENCE, so the reference's own analysis
                             scripts read the file with a plain `torch.load` (the `--reference_pickles` switch of the CLI).  With the
                             reference package importable its classes are used; without it the names are lent for the duration
                             of the save.
Like the reference's `torch.load`, both unpickle / write files the user names: load only files you trust.
"""
import importlib
import sys
import types
from collections import namedtuple

import torch

from .lightning_checkpoint import OWN_PACKAGE, REFERENCE_PACKAGE, _pickle_module


def load(path, map_location=None):
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_pickle_module)


class _Names:
    """Named tuples of this package -> the classes of the same name at the same relative path of the reference package: the
    reference's own when it can be imported, otherwise stand-ins registered under its module names for the duration of a save."""

    def __init__(self):
        self.classes, self.lent_modules = {}, []

    def target(self, cls):
        if cls not in self.classes:
            module_name = REFERENCE_PACKAGE + cls.__module__[len(OWN_PACKAGE):]
            try:
                found = getattr(importlib.import_module(module_name), cls.__name__)
            except (ImportError, AttributeError):
                # lend the name: pickle checks that `module.<name> is the class` when it writes the reference to it
                found = namedtuple(cls.__name__, cls._fields)
                found.__module__ = module_name
                parts = module_name.split(".")
                for k in range(1, len(parts) + 1):
                    name = ".".join(parts[:k])
                    if name not in sys.modules:
                        sys.modules[name] = types.ModuleType(name)
                        sys.modules[name].__path__ = []
                        self.lent_modules.append(name)
                setattr(sys.modules[module_name], cls.__name__, found)
            self.classes[cls] = found
        return self.classes[cls]

    def renamed(self, obj):
        if isinstance(obj, tuple) and hasattr(obj, "_fields") and type(obj).__module__.startswith(OWN_PACKAGE):
            return self.target(type(obj))(*(self.renamed(field) for field in obj))
        if isinstance(obj, dict):
            return {key: self.renamed(value) for key, value in obj.items()}
        if isinstance(obj, list):
            return [self.renamed(value) for value in obj]
        if type(obj) is tuple:
            return tuple(self.renamed(value) for value in obj)
        return obj

    def release(self):
        for name in self.lent_modules:
            del sys.modules[name]


def save_for_reference(obj, path):
    names = _Names()
    try:
        with open(path, "wb") as fd:
            torch.save(names.renamed(obj), fd)
    finally:
        names.release()
