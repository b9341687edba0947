"""Factories with the reference's names (mmpt/models/builder.py:9-70)."""
import torch.nn as nn

from .registry import BACKBONES, COMPONENTS, DROP_LAYERS, LOSSES, MODELS, OPERATORS, build_from_cfg


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_components(cfg):
    return build(cfg, COMPONENTS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_drop_layer(cfg):
    return build(cfg, DROP_LAYERS)


def build_operators(cfg):
    return build(cfg, OPERATORS)


def build_model(cfg, train_cfg=None, test_cfg=None):
    """train_cfg / test_cfg are injected as constructor kwargs (builder.py:36-44)."""
    return build(cfg, MODELS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
