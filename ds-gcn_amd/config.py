"""Python-file configs with ``_base_`` inheritance (mmcv-1.x ``Config`` semantics, restated:
the reference's configs are plain python files merged dict-wise over their ``_base_`` list —
e.g. configs/dsstgcn/ntu60_xsub_3dkp/j.py:1 -> ../DSSTGCN_model.py:1 -> ../_init_/lr_schedual.py).
Child keys override base keys; dicts merge recursively unless the child sets ``_delete_=True``."""
import copy
import os
import types

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'


class ConfigDict(dict):
    """dict with attribute access (missing key -> AttributeError, like mmcv's ConfigDict)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{self.__class__.__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


def _merge(child, base):
    """Returns base updated by child (mmcv Config._merge_a_into_b)."""
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get(DELETE_KEY, False):
            out[k] = _merge(v, out[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            out[k] = copy.deepcopy(v)
    return out


def _file2dict(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(f'file "{filename}" does not exist')
    if not filename.endswith('.py'):
        raise IOError('Only py type are supported now!')
    with open(filename, 'r', encoding='utf-8') as f:
        src = f.read()
    ns = {'__file__': filename}
    exec(compile(src, filename, 'exec'), ns)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    if BASE_KEY in cfg:
        bases = cfg.pop(BASE_KEY)
        bases = bases if isinstance(bases, (list, tuple)) else [bases]
        base_cfg = {}
        for b in bases:
            bd = _file2dict(os.path.join(os.path.dirname(filename), b))
            dup = base_cfg.keys() & bd.keys()
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {sorted(dup)}')
            base_cfg.update(bd)
        cfg = _merge(cfg, base_cfg)
    return cfg


class Config:
    """``Config.fromfile(path)`` / ``Config(dict)``; attribute and item access; ``merge_from_dict``."""

    def __init__(self, cfg_dict=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f'cfg_dict must be a dict, but got {type(cfg_dict)}')
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_file2dict(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    __setitem__ = __setattr__

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def items(self):
        return self._cfg_dict.items()

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))

    def merge_from_dict(self, options):
        """Dotted keys: ``{'model.backbone.graph_cfg.layout': 'coco'}``."""
        nested = {}
        for full, v in options.items():
            d = nested
            keys = full.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        object.__setattr__(self, '_cfg_dict', _wrap(_merge(nested, dict(self._cfg_dict))))

    def __repr__(self):
        return f'Config (path: {self._filename}): {dict(self._cfg_dict)!r}'
