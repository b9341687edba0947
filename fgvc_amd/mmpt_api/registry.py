"""Registries with the names and call pattern of the reference (mmpt/models/registry.py:4-10,
which instantiates mmcv.utils.Registry).  mmcv is not a dependency: this is a ~40-line registry."""
from __future__ import annotations

import inspect


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._module_dict = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._module_dict)

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._module_dict)})"

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        if not (inspect.isclass(cls) or callable(cls)):
            raise TypeError(f"module must be a class or callable, got {type(cls)}")
        for n in ([name] if isinstance(name, str) else (name or [cls.__name__])):
            if not force and n in self._module_dict:
                raise KeyError(f"{n} is already registered in {self._name}")
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls

        return deco


def build_from_cfg(cfg, registry: Registry, default_args=None):
    """cfg['type'] names a registered class; remaining keys (+ default_args) are its kwargs."""
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, got {type(cfg)}")
    if "type" not in cfg and not (default_args and "type" in default_args):
        raise KeyError(f"`cfg` or `default_args` must contain the key 'type': {cfg}")
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    typ = args.pop("type")
    if isinstance(typ, str):
        cls = registry.get(typ)
        if cls is None:
            raise KeyError(f"{typ} is not in the {registry.name} registry")
    elif inspect.isclass(typ) or callable(typ):
        cls = typ
    else:
        raise TypeError(f"type must be a str or class, got {type(typ)}")
    return cls(**args)


MODELS = Registry("model")
BACKBONES = Registry("backbone")
COMPONENTS = Registry("component")
OPERATORS = Registry("operators")
LOSSES = Registry("loss")
DROP_LAYERS = Registry("drop_layer")
