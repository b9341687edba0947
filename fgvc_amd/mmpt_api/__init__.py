"""The slice of the reference's `mmpt` package that the label-propagation inference path touches,
with the same names, backed by libfgvc_hip.so.

    import fgvc_amd; fgvc_amd.install_as_mmpt()
    from mmpt.models import build_model                     # same import lines as tools/test.py:12-15
    from mmpt.models.common import masked_attention_efficient, spatial_neighbor
"""
from . import backbones, builder, common, config, registry, trackers  # noqa: F401
from .backbones import ResNet, load_checkpoint  # noqa: F401
from .builder import (build, build_backbone, build_components, build_drop_layer, build_loss, build_model,  # noqa: F401
                      build_operators)
from .config import Config, ConfigDict  # noqa: F401
from .registry import BACKBONES, COMPONENTS, DROP_LAYERS, LOSSES, MODELS, OPERATORS, Registry, build_from_cfg  # noqa: F401
from .trackers import BaseModel, BaseTracker, HRVanillaTracker, VanillaTracker  # noqa: F401
