"""mmcv.Config look-alike for the reference's python-dict config files
(configs/eval/res18_d1_eval.py with `_base_ = './base_data.py'`)."""
from __future__ import annotations

import os
import types


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted on the way in."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return ConfigDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{k}'") from None

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]


def _merge(base: dict, child: dict) -> dict:
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
            out[k] = v
    return out


def _load(path: str) -> dict:
    path = os.path.abspath(path)
    with open(path) as f:
        src = f.read()
    ns = {"__file__": path}
    exec(compile(src, path, "exec"), ns)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    bases = cfg.pop("_base_", None)
    if bases is not None:
        merged = {}
        for b in ([bases] if isinstance(bases, str) else bases):
            merged = _merge(merged, _load(os.path.join(os.path.dirname(path), b)))
        cfg = _merge(merged, cfg)
    return cfg


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, "_cfg_dict", ConfigDict(cfg_dict or {}))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def fromfile(filename: str) -> "Config":
        return Config(_load(filename), filename)

    def __getattr__(self, k):
        return getattr(self._cfg_dict, k)

    def __setattr__(self, k, v):
        self._cfg_dict[k] = v

    def __getitem__(self, k):
        return self._cfg_dict[k]

    def __setitem__(self, k, v):
        self._cfg_dict[k] = v

    def __contains__(self, k):
        return k in self._cfg_dict

    def get(self, k, default=None):
        return self._cfg_dict.get(k, default)

    def merge_from_dict(self, options: dict):
        for k, v in options.items():
            d = self._cfg_dict
            parts = k.split(".")
            for p in parts[:-1]:
                d = d.setdefault(p, ConfigDict())
            d[parts[-1]] = v
