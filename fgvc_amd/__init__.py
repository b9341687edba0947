"""fgvc_amd: FGVC's label-propagation inference hot path, MI355X-native.

Product = fgvc_amd/lib/libfgvc_hip.so (hand-written HIP for gfx950, C ABI in include/fgvc_hip.h)
plus this thin Python host layer mirroring the reference's `mmpt` operator/registry interface.
There is no CPU fallback anywhere in this package.
"""
import sys
import types

__version__ = "0.1.0"


def install_as_mmpt(force: bool = False):
    """Register fgvc_amd.mmpt_api under the reference's module names (`mmpt`, `mmpt.models`,
    `mmpt.models.common`, ...) so code written against the reference imports unchanged."""
    from . import mmpt_api as api

    if "mmpt" in sys.modules and not force and not getattr(sys.modules["mmpt"], "_fgvc_amd", False):
        raise RuntimeError("another `mmpt` package is already imported")

    def mod(name, src=None, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m._fgvc_amd = True
        if src is not None:
            for k in dir(src):
                if not k.startswith("__"):
                    setattr(m, k, getattr(src, k))
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mmpt = mod("mmpt")
    models = mod("mmpt.models", api)
    mmpt.models = models
    for leaf, src in (("registry", api.registry), ("builder", api.builder), ("common", api.common),
                      ("backbones", api.backbones), ("trackers", api.trackers)):
        setattr(models, leaf, mod(f"mmpt.models.{leaf}", src))
    from . import apis
    mmpt.apis = mod("mmpt.apis", apis)
    return mmpt
