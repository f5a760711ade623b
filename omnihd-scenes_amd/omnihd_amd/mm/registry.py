"""A small ``Registry`` with mmcv's decorator/`build` semantics (``type='Name'`` dicts)."""
import inspect


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def __contains__(self, key):
        return key in self._modules

    def get(self, key):
        return self._modules.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            keys = [name] if isinstance(name, str) else (name or [cls.__name__])
            for k in keys:
                if k in self._modules and not force and self._modules[k] is not cls:
                    raise KeyError(f"{k} is already registered in {self.name}")
                self._modules[k] = cls
            return cls
        if module is not None:
            return _register(module)
        if callable(name) and not isinstance(name, str):      # used as bare @REG.register_module
            cls, name = name, None
            return _register(cls)
        return _register

    def build(self, cfg, **default_args):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if cfg is None:
        return None
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise TypeError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    t = args.pop("type")
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError(f"{t} is not in the {registry.name} registry")
    if not inspect.isclass(cls) and not callable(cls):
        raise TypeError(f"type must be a str or class, got {type(cls)}")
    return cls(**args)


DETECTORS = Registry("detector")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
VOXEL_ENCODERS = Registry("voxel_encoder")
MIDDLE_ENCODERS = Registry("middle_encoder")
NORM_LAYERS = Registry("norm layer")
PIPELINES = Registry("pipeline")      # mmdet.datasets.builder.PIPELINES: data-pipeline steps by ``type=``
