"""Model registry and builders.

Drop-in surface of the reference's ``pyskl/models/builder.py:5-43``: ONE registry object that
answers to five names (backbones, necks, heads, recognizers and losses all share it), a
``build_<kind>(cfg)`` helper per kind, and ``build_model(cfg)`` which only accepts registered
recognizer types and raises ``ValueError`` otherwise.
"""
from .registry import Registry

MODELS = Registry('models')
BACKBONES = NECKS = HEADS = RECOGNIZERS = LOSSES = MODELS


def _builder(kind):
    def build(cfg):
        return MODELS.build(cfg)
    build.__name__ = f'build_{kind}'
    build.__doc__ = f'Instantiate a {kind} from its config dict (``type`` + ctor kwargs).'
    return build


build_backbone = _builder('backbone')
build_neck = _builder('neck')
build_head = _builder('head')
build_recognizer = _builder('recognizer')
build_loss = _builder('loss')


def build_model(cfg):
    kind = dict(cfg).get('type')
    if kind is None or kind not in RECOGNIZERS:
        raise ValueError(f'{kind} is not registered')
    return build_recognizer(cfg)
