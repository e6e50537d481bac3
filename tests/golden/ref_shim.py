"""Import the read-only reference (/root/reference) with stub third-party modules.

Build-container only: the GPU box has no /root/reference.  Used by
``tests/golden/gen_golden.py`` (fixture generation) and by the optional
the ``*_live`` tests of ``tests/test_oracle_golden.py`` (skipped when the reference is absent).

Nothing from the reference is copied: the real modules are imported in place
(``sys.dont_write_bytecode`` so no __pycache__ lands in the read-only tree).
The stubs below replace *third-party* packages that are not installed here
(mmcv, cv2, tkinter, turtle, torch_geometric via heads/gread); their behaviour is
the documented mmcv-1.5 behaviour.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get('DSGCN_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'pyskl', 'models', 'gcns'))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _shell(name, relpath):
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(REF_ROOT, relpath)]
    m.__package__ = name
    sys.modules[name] = m
    return m


class _Registry:
    """mmcv.utils.Registry, reduced: name -> class, build(cfg) pops 'type'."""

    def __init__(self, name, parent=None, **kw):
        self.name = name
        self._module_dict = {}

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def build(self, cfg, *a, **kw):
        args = dict(cfg)
        typ = args.pop('type')
        cls = self._module_dict[typ] if isinstance(typ, str) else typ
        return cls(**args)


_loaded = None


def load():
    """Returns a namespace with the reference's hot-path modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError('reference tree not present at ' + REF_ROOT)
    sys.dont_write_bytecode = True
    import torch
    import torch.nn as nn

    def build_norm_layer(cfg, num_features, postfix=''):
        cfg = dict(cfg)
        typ = cfg.pop('type')
        assert typ in ('BN', 'BN2d'), typ
        cfg.setdefault('eps', 1e-5)
        return 'bn' + str(postfix), nn.BatchNorm2d(num_features, **cfg)

    def build_activation_layer(cfg):
        cfg = dict(cfg)
        typ = cfg.pop('type')
        return {'ReLU': nn.ReLU, 'Tanh': nn.Tanh, 'Sigmoid': nn.Sigmoid}[typ](**cfg)

    def normal_init(module, mean=0, std=1, bias=0):
        nn.init.normal_(module.weight, mean, std)
        if getattr(module, 'bias', None) is not None:
            nn.init.constant_(module.bias, bias)

    _mod('tkinter', N='n')
    _mod('turtle', screensize=None)
    _mod('cv2', KeyPoint=None, threshold=None)
    _mod('mmcv', __version__='1.5.0')
    _mod('mmcv.cnn', MODELS=_Registry('model'), build_norm_layer=build_norm_layer,
         build_activation_layer=build_activation_layer, normal_init=normal_init, ConvModule=None)
    _mod('mmcv.utils', Registry=_Registry, _BatchNorm=nn.modules.batchnorm._BatchNorm)
    _mod('mmcv.runner', load_checkpoint=lambda *a, **k: None, DistEvalHook=object)

    _shell('pyskl', 'pyskl')
    _shell('pyskl.models', 'pyskl/models')
    _shell('pyskl.models.gcns', 'pyskl/models/gcns')
    _shell('pyskl.models.heads', 'pyskl/models/heads')
    _shell('pyskl.models.recognizers', 'pyskl/models/recognizers')
    _shell('pyskl.models.losses', 'pyskl/models/losses')
    utils = _shell('pyskl.utils', 'pyskl/utils')
    core = _shell('pyskl.core', 'pyskl/core')
    _mod('pyskl.models.heads.gread', global_add_pool=None, global_mean_pool=None,
         global_max_pool=None, GlobalAttention=None, Set2Set=None)

    graph = importlib.import_module('pyskl.utils.graph')
    utils.Graph = graph.Graph
    utils.cache_checkpoint = lambda x: x
    evaluation = None
    try:
        evaluation = importlib.import_module('pyskl.core.evaluation')
        core.top_k_accuracy = evaluation.top_k_accuracy
    except Exception:  # sklearn / mmcv eval hook imports may fail; restate the tiny fn use
        import numpy as np

        def top_k_accuracy(scores, labels, topk=(1,)):
            res = []
            labels = np.array(labels)[:, np.newaxis]
            for k in topk:
                pred = np.argsort(scores, axis=1)[:, -k:][:, ::-1]
                m = np.logical_or.reduce(pred == labels, axis=1)
                res.append(m.sum() / m.shape[0])
            return res
        core.top_k_accuracy = top_k_accuracy

    builder = importlib.import_module('pyskl.models.builder')
    sys.modules['pyskl.models'].builder = builder
    gutils = importlib.import_module('pyskl.models.gcns.utils')
    dgstgcn = importlib.import_module('pyskl.models.gcns.dgstgcn')
    stgcn = importlib.import_module('pyskl.models.gcns.stgcn')
    ctrgcn = importlib.import_module('pyskl.models.gcns.ctrgcn')
    aagcn = importlib.import_module('pyskl.models.gcns.aagcn')
    ce = importlib.import_module('pyskl.models.losses.cross_entropy_loss')
    head = importlib.import_module('pyskl.models.heads.simple_head')
    rec = importlib.import_module('pyskl.models.recognizers.recognizergcn')
    _loaded = types.SimpleNamespace(
        graph=graph, builder=builder, gutils=gutils, dgstgcn=dgstgcn, stgcn=stgcn, ctrgcn=ctrgcn, aagcn=aagcn,
        ce=ce, head=head, rec=rec, evaluation=evaluation, torch=torch)
    return _loaded
