"""Config surface: python-file configs with attribute access, as the reference's
Config.fromfile / ConfigDict (minddet/models/centerpoint/det3d_ms/torchie/utils/config.py:53-162),
plus a flat YAML loader with --config_path style CLI overrides
(minddet/models/centernet/src/model_utils/config.py:147-171)."""
import argparse
import os
import runpy


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            v = self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '{}'".format(name))
        return v

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


class Config(object):
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, "_cfg_dict", _wrap(cfg_dict or {}))
        object.__setattr__(self, "_filename", filename)

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        if filename.endswith(".py"):
            ns = runpy.run_path(filename)
            cfg = {k: v for k, v in ns.items() if not k.startswith("__") and not callable(v)
                   and not isinstance(v, type(os))}
        elif filename.endswith((".yml", ".yaml", ".json")):
            import yaml

            with open(filename) as f:
                cfg = yaml.load(f, Loader=yaml.SafeLoader)
        else:
            raise IOError("Only py/yml/yaml/json type are supported now!")
        return Config(cfg, filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __repr__(self):
        return "Config (path: {}): {}".format(self._filename, dict.__repr__(self._cfg_dict))


def get_config(argv=None, default_path=None):
    """`--config_path=...` + automatic CLI overrides for flat keys (centernet config.py:147-171)."""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--config_path", type=str, default=default_path)
    known, rest = pre.parse_known_args(argv)
    cfg = Config.fromfile(known.config_path)
    parser = argparse.ArgumentParser(parents=[pre])
    for k, v in cfg._cfg_dict.items():
        if isinstance(v, (int, float, str, bool)):
            t = (lambda s: s.lower() in ("1", "true", "yes")) if isinstance(v, bool) else type(v)
            parser.add_argument("--" + k, type=t, default=v)
    args = parser.parse_args(argv)
    for k, v in vars(args).items():
        if k != "config_path":
            cfg._cfg_dict[k] = v
    return cfg
