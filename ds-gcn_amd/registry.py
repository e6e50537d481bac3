"""Registry / build_from_cfg with the mmcv-1.x semantics the reference relies on
(reference: pyskl/models/builder.py:5-10 builds ``Registry('models', parent=MMCV_MODELS)``;
mmcv-full==1.5.0 is third-party and absent here, so its documented behaviour is restated:
``register_module()`` decorator keyed by class name, ``build(cfg)`` pops ``type`` and calls
the class with the remaining keys, unknown ``type`` raises ``KeyError``)."""
import inspect


class Registry:

    def __init__(self, name, parent=None):
        self._name = name
        self._module_dict = {}
        self.parent = parent

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{self.__class__.__name__}(name={self._name}, items={list(self._module_dict)})'

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f'module must be a class, but got {type(cls)}')
        names = [name or cls.__name__] if not isinstance(name, (list, tuple)) else list(name)
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self._name}')
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}')
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    return obj_cls(**args)
