"""Model-build surface: same names and behaviour as the reference's registry pattern
(minddet/models/centerpoint/det3d_ms/utils/registry.py:6-78, models/registry.py:3-10,
models/builder.py:16-53): Registry(name), @REG.register_module, build_from_cfg(cfg{"type":...}),
build_detector(cfg.model, train_cfg, test_cfg)."""
import inspect


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    def __repr__(self):
        return "{}(name={}, items={})".format(self.__class__.__name__, self._name, list(self._module_dict))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError("module must be a class, but got {}".format(type(cls)))
        if cls.__name__ in self._module_dict:
            raise KeyError("{} is already registered in {}".format(cls.__name__, self.name))
        self._module_dict[cls.__name__] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    if not (isinstance(cfg, dict) and "type" in cfg):
        raise AssertionError('cfg must be a dict containing the key "type"')
    if not (isinstance(default_args, dict) or default_args is None):
        raise AssertionError("default_args must be a dict or None")
    args = dict(cfg)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError("{} is not in the {} registry".format(obj_type, registry.name))
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError("type must be a str or valid type, but got {}".format(type(obj_type)))
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
    return obj_cls(**args)


READERS = Registry("reader")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")
SECOND_STAGE = Registry("second_stage")
ROI_HEAD = Registry("roi_head")


def build(cfg, registry, default_args=None):
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
