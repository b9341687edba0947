"""Load the reference's hot-path files from /root/reference WITHOUT mmcv.

TEST INFRASTRUCTURE, BUILD CONTAINER ONLY.  Nothing here is shipped to the GPU
box as reference code: this module reads the reference *in place* (read-only)
so that `tests/golden/gen_golden.py` can record the input/output vectors that pin the CPU
restatement in `oracle/fgvc_oracle.py` (`tests/test_oracle.py`).  It is never imported by the product package.

The reference package (`import mmpt`) needs mmcv-full==1.5.2, cv2, av, ... which
are absent here.  The hot-path *files* however only touch a
handful of mmcv symbols; this loader installs a minimal stand-in for those
symbols (plain torch.nn arithmetic: Conv2d + BatchNorm2d + ReLU), registers an
empty `mmpt` package skeleton in sys.modules and executes the genuine files
under it.

Files executed (read from /root/reference, never copied):
    mmpt/models/registry.py, mmpt/models/builder.py
    mmpt/models/common/{utils,affinity_utils,corr_lookup,part_unfold,
                        local_attention,correlation}.py
    mmpt/models/backbones/resnet.py
    mmpt/models/trackers/{base,vanilla_tracker}.py
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("FGVC_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "mmpt/models/common/local_attention.py"))


# ----------------------------------------------------------------------------
# mmcv stand-in (only the symbols the hot-path files touch)
# ----------------------------------------------------------------------------
class _ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def get(self, key):
        return self.module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls

        if module is not None:
            return deco(module)
        return deco


def _build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop("type")
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f"{typ} is not in the {registry.name} registry")
    return cls(**args)


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        for m in self.children():
            if hasattr(m, "init_weights"):
                m.init_weights()


class _ConvModule(nn.Module):
    """conv -> BN -> ReLU, attribute names `conv`, `bn`, `activate` as in mmcv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, groups=1, bias="auto", conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), inplace=True, **kw):
        super().__init__()
        with_norm = norm_cfg is not None
        if bias == "auto":
            bias = not with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels) if with_norm else None
        self.activate = nn.ReLU(inplace=inplace) if act_cfg is not None else None

    @property
    def norm(self):
        return self.bn

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.activate is not None:
            x = self.activate(x)
        return x


def _kaiming_init(module, a=0, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
    nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def _constant_init(module, val, bias=0):
    if getattr(module, "weight", None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


class _Correlation(nn.Module):
    """Stand-in for mmcv.ops.Correlation (mmcv-full==1.5.2, CUDA only, absent here) with the arguments
    HRVanillaTracker passes (vanilla_tracker.py:426-428): kernel_size=1, stride=1, padding=0, dilation_patch=1.
    out[b, ph, pw, y, x] = sum_c in1[b,c,y,x] * in2[b,c,y+ph-R,x+pw-R], zero outside the map -- restated from mmcv's
    published semantics and from the reference's own torch-only twin (local_attention.py:1190-1198); `dilation` is the
    dilation of the correlation KERNEL, which has one tap here, so it changes nothing.  The arithmetic of this class is
    "parity unpinned"; what the HR golden pins is the reference's DRIVER LOOP around it."""

    def __init__(self, kernel_size=1, max_displacement=1, stride=1, padding=0, dilation=1, dilation_patch=1):
        super().__init__()
        assert kernel_size == 1 and stride == 1 and padding == 0 and dilation_patch == 1
        self.R = max_displacement

    def forward(self, input1, input2):
        import torch.nn.functional as F
        B, C, H, W = input1.shape
        L = 2 * self.R + 1
        unf = F.unfold(input2, kernel_size=L, padding=self.R).reshape(B, C, L * L, H, W)
        return (unf * input1.unsqueeze(2)).sum(1).reshape(B, L, L, H, W)


def _auto_fp16(*a, **k):
    def deco(fn):
        return fn
    return deco


def _install_mmcv_stub():
    if "mmcv" in sys.modules and getattr(sys.modules["mmcv"], "_fgvc_stub", False):
        return
    mmcv = types.ModuleType("mmcv")
    mmcv._fgvc_stub = True
    mmcv.build_from_cfg = _build_from_cfg
    mmcv.ConfigDict = _ConfigDict
    cnn = types.ModuleType("mmcv.cnn")
    cnn.ConvModule = _ConvModule
    cnn.kaiming_init = _kaiming_init
    cnn.constant_init = _constant_init
    runner = types.ModuleType("mmcv.runner")
    runner.BaseModule = _BaseModule
    runner.auto_fp16 = _auto_fp16
    runner.load_checkpoint = lambda *a, **k: None
    runner._load_checkpoint = lambda *a, **k: {}
    utils = types.ModuleType("mmcv.utils")
    utils.Registry = _Registry
    utils._BatchNorm = nn.modules.batchnorm._BatchNorm
    utils.ConfigDict = _ConfigDict
    mops = types.ModuleType("mmcv.ops")
    mops.Correlation = _Correlation
    mmcv.cnn, mmcv.runner, mmcv.utils, mmcv.ops = cnn, runner, utils, mops
    sys.modules.update({"mmcv": mmcv, "mmcv.cnn": cnn, "mmcv.runner": runner, "mmcv.utils": utils, "mmcv.ops": mops})
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except Exception:  # pragma: no cover
            sys.modules["tqdm"] = types.ModuleType("tqdm")


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []  # mark as package
    sys.modules[name] = m
    return m


def _exec(modname, relpath):
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    parent, _, leaf = modname.rpartition(".")
    setattr(sys.modules[parent], leaf, mod)
    return mod


def _star(dst, src):
    names = getattr(src, "__all__", None) or [n for n in vars(src) if not n.startswith("_")]
    for n in names:
        setattr(dst, n, getattr(src, n))


_LOADED = None


def load():
    """Returns a namespace with the genuine reference callables."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    _install_mmcv_stub()
    if "mmpt" in sys.modules and not getattr(sys.modules["mmpt"], "_fgvc_ref", False):
        raise RuntimeError("a different `mmpt` is already imported (fgvc_amd.install_as_mmpt?); "
                           "load the reference in a separate process")
    mmpt = _pkg("mmpt"); mmpt._fgvc_ref = True
    models = _pkg("mmpt.models"); mmpt.models = models
    common = _pkg("mmpt.models.common"); models.common = common
    backbones = _pkg("mmpt.models.backbones"); models.backbones = backbones
    trackers = _pkg("mmpt.models.trackers"); models.trackers = trackers
    utils_pkg = _pkg("mmpt.utils"); mmpt.utils = utils_pkg
    utils_pkg.get_root_logger = lambda *a, **k: __import__("logging").getLogger("mmpt")
    utils_pkg.__all__ = []  # `from ...utils import *` in the tracker brings nothing it needs

    _exec("mmpt.models.registry", "mmpt/models/registry.py")
    _exec("mmpt.models.builder", "mmpt/models/builder.py")
    mods = {}
    for leaf in ("utils", "affinity_utils", "corr_lookup", "part_unfold", "local_attention", "correlation"):
        if leaf == "local_attention":
            # In the real package local_attention.py:9 runs while common/__init__ is still
            # half-initialised, so `part_unfold` binds to the sub-MODULE (:1195 calls part_unfold.part_unfold).
            common.part_unfold = mods["part_unfold"]
        mod = _exec(f"mmpt.models.common.{leaf}", f"mmpt/models/common/{leaf}.py")
        mods[leaf] = mod
        _star(common, mod)  # NB: correlation.py defines a *function* local_attention that shadows the module
    common.part_unfold = mods["part_unfold"]  # local_attention.py:9 uses the module, :1195 calls part_unfold.part_unfold
    resnet = _exec("mmpt.models.backbones.resnet", "mmpt/models/backbones/resnet.py")
    backbones.ResNet = resnet.ResNet
    # names the tracker imports from ..common that live in files we did not load
    for missing in ("masked_attention_efficient_correlation",):
        if not hasattr(common, missing):
            setattr(common, missing, None)
    _exec("mmpt.models.trackers.base", "mmpt/models/trackers/base.py")
    # HRVanillaTracker imports mmcv.ops lazily in __init__ (the _Correlation stand-in above); VanillaTracker does not need it.
    vt = _exec("mmpt.models.trackers.vanilla_tracker", "mmpt/models/trackers/vanilla_tracker.py")

    ns = types.SimpleNamespace(
        common=common,
        masked_attention_efficient=mods['local_attention'].masked_attention_efficient,
        masked_attention_efficient_v2=mods['local_attention'].masked_attention_efficient_v2,
        masked_attention_efficient_c2f=mods['local_attention'].masked_attention_efficient_c2f,
        masked_attention_efficient_correlation_v2=mods['local_attention'].masked_attention_efficient_correlation_v2,
        masked_attention=mods['local_attention'].masked_attention,
        local_square_attention=mods['local_attention'].local_square_attention,
        spatial_neighbor=mods['affinity_utils'].spatial_neighbor,
        compute_affinity=mods['affinity_utils'].compute_affinity,
        propagate=mods['affinity_utils'].propagate,
        non_local_attention=mods['correlation'].non_local_attention,
        coords_grid=mods['local_attention'].coords_grid,
        ResNet=resnet.ResNet,
        VanillaTracker=vt.VanillaTracker,
        HRVanillaTracker=vt.HRVanillaTracker,
        builder=sys.modules["mmpt.models.builder"],
        registry=sys.modules["mmpt.models.registry"],
        ConfigDict=_ConfigDict,
    )
    _LOADED = ns
    return ns


class cuda_as_cpu:
    """Context manager: the reference driver hard-codes `.cuda()`
    (vanilla_tracker.py:194-195,251-255,284-287,407,409); make it the identity."""

    def __enter__(self):
        self._t = torch.Tensor.cuda
        self._m = nn.Module.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        nn.Module.cuda = lambda self, *a, **k: self
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda = self._t
        nn.Module.cuda = self._m
        return False
