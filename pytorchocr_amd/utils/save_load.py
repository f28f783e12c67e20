"""Checkpoint loading with the semantics of reference pytocr/utils/save_load.py:81-101: the file may be a bare state_dict
or {"state_dict": ...}; a DataParallel/DDP `module.` prefix is tolerated in either direction; the merged dict is loaded with
strict=True, so a key that matches nothing raises (and keys the checkpoint lacks keep the model's own values)."""
import os

import torch


def _target_key(key, model_keys):
    """name under which checkpoint entry `key` goes into the model's state_dict"""
    if key in model_keys:
        return key
    prefixed = "module." + key
    if prefixed in model_keys:
        return prefixed
    return key.replace("module.", "")


def load_pretrained_params(model, path):
    if not os.path.exists(path):
        raise AssertionError("The {} does not exists!".format(path))
    ckpt = torch.load(path, map_location="cpu")
    ckpt = ckpt.get("state_dict", ckpt)
    merged = model.state_dict()
    model_keys = set(merged.keys())
    merged.update({_target_key(k, model_keys): v for k, v in ckpt.items()})
    model.load_state_dict(merged, strict=True)
    return model
