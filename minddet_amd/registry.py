"""Model-build surface.  The NAMES are the reference's, because configs and callers use them
(minddet/models/centerpoint/det3d_ms/utils/registry.py:6-78, models/registry.py:3-10, models/builder.py:16-53):
`Registry(name)` with `.name`, `.module_dict`, `.get(key)`, `@REG.register_module`; `build_from_cfg(cfg, registry, default_args)`
where `cfg["type"]` names a registered class (or is the class) and the remaining keys are its constructor arguments, `default_args`
filling the ones cfg leaves out; `build_detector(cfg.model, train_cfg, test_cfg)`.  Everything behind those names is this build's own:
a name -> class table, one resolver, error texts that say what to do."""
import inspect


class Registry:
    """A table from class name to class for one kind of component."""

    def __init__(self, name):
        self._kind = str(name)
        self._table = {}

    @property
    def name(self):
        return self._kind

    @property
    def module_dict(self):
        return self._table

    def get(self, key):
        return self._table.get(key)

    def __contains__(self, key):
        return key in self._table

    def __len__(self):
        return len(self._table)

    def __repr__(self):
        return f"<{self._kind} registry: {', '.join(sorted(self._table)) or 'empty'}>"

    def register_module(self, cls):
        """class decorator; the class is registered under its own name and returned unchanged"""
        if not inspect.isclass(cls):
            raise TypeError(f"{self._kind} registry: only classes can be registered, got an object of type {type(cls).__name__}")
        known = self._table.get(cls.__name__)
        if known is not None:
            raise KeyError(f"{self._kind} registry: the name {cls.__name__!r} is taken by {known.__module__}.{known.__qualname__}")
        self._table[cls.__name__] = cls
        return cls


def _resolve(kind, registry):
    """cfg["type"] -> class: a registered name, or a class given directly"""
    if inspect.isclass(kind):
        return kind
    if not isinstance(kind, str):
        raise TypeError(f'"type" has to be a registered name or a class, not {type(kind).__name__}')
    cls = registry.get(kind)
    if cls is None:
        raise KeyError(f"no {registry.name} called {kind!r}; registered: {sorted(registry.module_dict)}")
    return cls


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise AssertionError(f'a {registry.name} config is a dict with a "type" entry, got {cfg!r}')
    if default_args is not None and not isinstance(default_args, dict):
        raise AssertionError(f"default_args is a dict of constructor arguments (or None), got {type(default_args).__name__}")
    kwargs = {k: v for k, v in cfg.items() if k != "type"}
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return _resolve(cfg["type"], registry)(**kwargs)


READERS = Registry("reader")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")
SECOND_STAGE = Registry("second_stage")
ROI_HEAD = Registry("roi_head")


def build(cfg, registry, default_args=None):
    """one config -> one object, a list of configs -> a list of objects"""
    if isinstance(cfg, list):
        return [build_from_cfg(c, registry, default_args) for c in cfg]
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_roi_head(cfg):
    return build(cfg, ROI_HEAD)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
