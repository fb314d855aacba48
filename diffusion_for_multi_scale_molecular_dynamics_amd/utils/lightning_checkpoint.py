"""Reading a checkpoint written by the reference's trainer -- without Lightning and without the reference package installed.

The reference's sampling script takes its network from a Lightning checkpoint (src/.../sample_diffusion.py:191-205:
`AXLDiffusionLightningModel.load_from_checkpoint(path).axl_network`): the file carries the weights (`state_dict`, keys
`axl_network.*`) AND, under `hyper_parameters`, the pickled `AXLDiffusionParameters` the model was built from
(models/axl_diffusion_lightning_model.py:62-95: `save_hyperparameters`), whose `score_network_parameters` describe the network.
A sampling configuration therefore holds no `model:` block (experiments/.../Si_1x1x1/config_sample_T=1000.yaml).

Unpickling that object needs the classes it names.  Here every class of the reference package is looked up under the SAME
relative module path in this package (the score-network parameter dataclasses live at the same paths with the same fields), and
anything that cannot be found -- Lightning's, the optimiser's, the loss parameters' classes: nothing the sampling path reads --
becomes an inert placeholder that keeps its state.  Like `torch.load(weights_only=False)` in the reference, this unpickles a file
the user supplies: load only checkpoints you trust.
"""
import dataclasses
import importlib
import pickle
import types
from typing import Any, Dict, Optional

import torch

REFERENCE_PACKAGE = "diffusion_for_multi_scale_molecular_dynamics"
OWN_PACKAGE = __name__.rsplit(".", 2)[0]


class Placeholder:
    """Stands in for a class this process does not have: built from anything, keeps what it is given."""

    def __init__(self, *args, **kwargs):
        self.placeholder_args, self.placeholder_kwargs = args, kwargs

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.placeholder_state = state

    def __call__(self, *args, **kwargs):         # (a pickled function or bound factory that is called while unpickling)
        return Placeholder(*args, **kwargs)


def _placeholder(module: str, name: str):
    return type(name, (Placeholder,), {"__module__": module, "placeholder_for": f"{module}.{name}"})


class TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == REFERENCE_PACKAGE or module.startswith(REFERENCE_PACKAGE + "."):
            for candidate in (OWN_PACKAGE + module[len(REFERENCE_PACKAGE):], module):
                try:
                    found = getattr(importlib.import_module(candidate), name)
                except (ImportError, AttributeError):
                    continue
                return found
            return _placeholder(module, name)
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return _placeholder(module, name)


# what torch.load expects of `pickle_module`
_pickle_module = types.SimpleNamespace(__name__="pickle", Unpickler=TolerantUnpickler, load=lambda f, **kw: TolerantUnpickler(f, **kw).load(),
                                       loads=pickle.loads, dump=pickle.dump, dumps=pickle.dumps)


def load_checkpoint(path) -> Dict[str, Any]:
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_pickle_module)


def score_network_parameters_of(checkpoint: Dict[str, Any]) -> Optional[Any]:
    """The `score_network_parameters` the checkpoint's model was built from, as THIS package's dataclass (rebuilt through its
    constructor: fields this package adds take their defaults), or None when the file holds no hyper-parameters."""
    from ..models.score_networks.score_network_factory import SCORE_NETWORK_PARAMETERS_BY_ARCH
    hyper = checkpoint.get("hyper_parameters")
    if hyper is None:
        return None
    holder = hyper.get("hyper_params", hyper) if isinstance(hyper, dict) else hyper
    stored = holder.get("score_network_parameters") if isinstance(holder, dict) else getattr(holder, "score_network_parameters", None)
    if stored is None:
        return None
    read = (lambda key: stored.get(key)) if isinstance(stored, dict) else (lambda key: getattr(stored, key, None))
    has = (lambda key: key in stored) if isinstance(stored, dict) else (lambda key: hasattr(stored, key))
    architecture = read("architecture")
    assert architecture in SCORE_NETWORK_PARAMETERS_BY_ARCH, \
        f"the checkpoint's score network has architecture {architecture!r}: not implemented here " \
        f"(choices: {list(SCORE_NETWORK_PARAMETERS_BY_ARCH)})"
    cls = SCORE_NETWORK_PARAMETERS_BY_ARCH[architecture]
    return cls(**{f.name: read(f.name) for f in dataclasses.fields(cls) if f.init and has(f.name)})


def state_dict_of(checkpoint: Dict[str, Any], prefix: str = "axl_network.") -> Dict[str, torch.Tensor]:
    """The score network's weights: Lightning's `state_dict` with the module prefix removed (a bare state_dict passes through)."""
    state = checkpoint.get("state_dict", checkpoint)
    return {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)} or state
