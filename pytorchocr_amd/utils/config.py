"""YAML config loading: mirror of reference deploy/utils.py:8-62 (AttrDict, load_config, merge_config with dotted keys)."""
import os

import yaml


class AttrDict(dict):
    """Single level attribute dict, NOT recursive"""

    def __init__(self, **kwargs):
        super(AttrDict, self).__init__()
        super(AttrDict, self).update(kwargs)

    def __getattr__(self, key):
        if key in self:
            return self[key]
        raise AttributeError("object has no attribute '{}'".format(key))


def merge_config(config, global_config):
    for key, value in config.items():
        if "." not in key:
            if isinstance(value, dict) and key in global_config:
                global_config[key].update(value)
            else:
                global_config[key] = value
        else:
            sub_keys = key.split(".")
            assert sub_keys[0] in global_config, \
                "the sub_keys can only be one of global_config: {}, but get: {}, please check your running command".format(
                    global_config.keys(), sub_keys[0])
            cur = global_config[sub_keys[0]]
            for idx, sub_key in enumerate(sub_keys[1:]):
                if idx == len(sub_keys) - 2:
                    cur[sub_key] = value
                else:
                    cur = cur[sub_key]
    return global_config


def load_config(file_path):
    """Load config from yml/yaml file (yaml.Loader: the reference's ymls use !!python/tuple, det_r18_db.yml:50)."""
    global_config = AttrDict()
    merge_config({"Global": {"debug": False}}, global_config)
    _, ext = os.path.splitext(file_path)
    assert ext in [".yml", ".yaml"], "only support yaml files for now"
    with open(file_path, "rb") as f:
        merge_config(yaml.load(f, Loader=yaml.Loader), global_config)
    return global_config
