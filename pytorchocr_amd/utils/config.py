"""YAML config loading with the semantics of the reference's deploy/utils.py:8-62: `load_config(path)` returns an
attribute-style top-level dict seeded with Global.debug = False; `merge_config(overrides, cfg)` merges one level deep
for plain keys and walks dotted keys ("Global.use_gpu") down existing sections."""
import os

import yaml


class AttrDict(dict):
    """dict whose TOP-LEVEL keys can also be read as attributes (nested dicts stay plain dicts, as in the reference)."""

    def __init__(self, **kwargs):
        dict.__init__(self, **kwargs)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("object has no attribute '{}'".format(name)) from None


def _set_dotted(cfg, dotted, value):
    head, *rest = dotted.split(".")
    if head not in cfg:
        raise AssertionError("the sub_keys can only be one of global_config: {}, but get: {}, "
                             "please check your running command".format(cfg.keys(), head))
    node = cfg[head]
    for part in rest[:-1]:
        node = node[part]
    node[rest[-1]] = value


def merge_config(config, global_config):
    """Merge `config` into `global_config` in place and return it.  A dict value under an existing plain key updates that
    section (one level); anything else replaces; "A.b.c" keys assign into the existing section A."""
    for key, value in config.items():
        if "." in key:
            _set_dotted(global_config, key, value)
        elif isinstance(value, dict) and key in global_config:
            global_config[key].update(value)
        else:
            global_config[key] = value
    return global_config


def load_config(file_path):
    """Read a .yml / .yaml config.  yaml.Loader (not safe_load): the reference's ymls carry python tags such as
    !!python/tuple (det_r18_db.yml:50)."""
    if os.path.splitext(file_path)[1] not in (".yml", ".yaml"):
        raise AssertionError("only support yaml files for now")
    cfg = AttrDict(Global={"debug": False})
    with open(file_path, "rb") as f:
        loaded = yaml.load(f, Loader=yaml.Loader)
    return merge_config(loaded, cfg)
