"""Import the reference's skeleton DATA PIPELINE (pyskl/datasets/pipelines) with stub third-party modules — build
container only, used by gen_golden_pipeline.py.  Nothing is copied: sampling.py, pose_related.py, formatting.py and
compose.py are imported in place; their heavy siblings (causal discovery, plotting, Neural_GC) and mmcv are stubbed."""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get('DSGCN_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'pyskl', 'datasets', 'pipelines'))


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
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self.module_dict.get(key)


def _build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    typ = args.pop('type')
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    return cls(**args)


_loaded = None


def load():
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError('reference tree not present at ' + REF_ROOT)
    sys.dont_write_bytecode = True
    saved = {k: sys.modules.get(k) for k in ('mmcv', 'mmcv.utils', 'mmcv.parallel')}
    _mod('mmcv', __version__='1.5.0', is_str=lambda x: isinstance(x, str))
    _mod('mmcv.utils', build_from_cfg=_build_from_cfg, Registry=_Registry)
    _mod('mmcv.parallel', DataContainer=lambda x, **kw: x)
    pipelines = _Registry('pipeline')
    if 'pyskl' not in sys.modules:
        _shell('pyskl', 'pyskl')
    _shell('pyskl.datasets', 'pyskl/datasets')
    _mod('pyskl.datasets.builder', PIPELINES=pipelines, DATASETS=_Registry('dataset'))
    _shell('pyskl.datasets.pipelines', 'pyskl/datasets/pipelines')
    _mod('pyskl.datasets.pipelines.causal')
    _mod('pyskl.datasets.pipelines.plot_confusion_metric')
    _shell('pyskl.datasets.pipelines.Neural_GC_master', 'pyskl/datasets/pipelines/Neural_GC_master')
    _shell('pyskl.datasets.pipelines.Neural_GC_master.models', 'pyskl/datasets/pipelines/Neural_GC_master/models')
    _mod('pyskl.datasets.pipelines.Neural_GC_master.models.clstm', cLSTM=None, train_model_ista=None)
    mods = {n: importlib.import_module('pyskl.datasets.pipelines.' + n)
            for n in ('compose', 'formatting', 'sampling', 'pose_related', 'augmentations')}
    for k, v in saved.items():            # leave the model-side stubs of ref_shim.py as they were
        if v is not None:
            sys.modules[k] = v
    _loaded = types.SimpleNamespace(PIPELINES=pipelines, Compose=mods['compose'].Compose, **mods)
    return _loaded
