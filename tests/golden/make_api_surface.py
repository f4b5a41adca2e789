"""Writes tests/golden/api_surface.json: the PUBLIC Python surface of the reference's MimiModel, LMModel and LMGen (the drop-in
boundary of SURVEY.md 8b) - every public attribute of the class that torch.nn.Module does not already have, with the parameter
names of the callables - obtained by importing the reference in this container:

    cd /tmp && PYTHONPATH=/root/reference/moshi NO_TORCH_COMPILE=1 PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_api_surface.py

tests/test_api_surface.py holds moshi_amd's classes to it (the file travels; the reference does not)."""
import inspect
import json
from pathlib import Path

import torch.nn as nn
from moshi.models.compression import MimiModel
from moshi.models.lm import LMGen, LMModel


def surface(cls):
    base = set(dir(nn.Module))
    out = {}
    for name in sorted(n for n in dir(cls) if not n.startswith("_") and n not in base):
        attr = inspect.getattr_static(cls, name)
        if isinstance(attr, property):
            out[name] = {"kind": "property"}
        elif callable(getattr(cls, name)):
            try:
                params = [p for p in inspect.signature(getattr(cls, name)).parameters if p != "self"]
            except (TypeError, ValueError):
                params = None
            out[name] = {"kind": "method", "params": params}
        else:
            out[name] = {"kind": "attribute"}
    out["__init__"] = {"kind": "method", "params": [p for p in inspect.signature(cls.__init__).parameters if p != "self"]}
    return out


def module_surface(mod):
    """Public names DEFINED in a module (functions with their parameters, classes with their public methods, constants)."""
    out = {}
    for name in sorted(n for n in dir(mod) if not n.startswith("_")):
        obj = getattr(mod, name)
        if inspect.isclass(obj) and obj.__module__ == mod.__name__:
            methods = {}
            for m in sorted(x for x in dir(obj) if not x.startswith("_") and callable(getattr(obj, x))):
                try:
                    methods[m] = [q for q in inspect.signature(getattr(obj, m)).parameters if q != "self"]
                except (TypeError, ValueError):
                    methods[m] = None
            out[name] = {"kind": "class", "methods": methods}
        elif inspect.isfunction(obj) and obj.__module__ == mod.__name__:
            out[name] = {"kind": "function", "params": list(inspect.signature(obj).parameters)}
        elif name.isupper() and isinstance(obj, (int, float, str)):
            out[name] = {"kind": "constant", "value": obj}
    return out


if __name__ == "__main__":
    from moshi.models import loaders
    res = {c.__name__: surface(c) for c in (MimiModel, LMModel, LMGen)}
    res["module:loaders"] = module_surface(loaders)
    path = Path(__file__).resolve().parent / "api_surface.json"
    path.write_text(json.dumps(res, indent=1, sort_keys=True) + "\n")
    print(path, {k: len(v) for k, v in res.items()})
