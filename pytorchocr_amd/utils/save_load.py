"""Checkpoint loading: mirror of reference pytocr/utils/save_load.py:81-101 (`module.` prefix tolerated both ways,
optional {"state_dict": ...} wrapper, strict=True)."""
import os

import torch


def load_pretrained_params(model, path):
    assert os.path.exists(path), "The {} does not exists!".format(path)
    pretrained_state_dict = torch.load(path, map_location="cpu")
    if "state_dict" in pretrained_state_dict:
        pretrained_state_dict = pretrained_state_dict["state_dict"]
    model_state_dict = model.state_dict()
    for k, v in pretrained_state_dict.items():
        if k in model_state_dict:
            name = k
        elif "module." + k in model_state_dict:
            name = "module." + k
        else:
            name = k.replace("module.", "")
        model_state_dict[name] = v
    model.load_state_dict(model_state_dict, strict=True)
    return model
