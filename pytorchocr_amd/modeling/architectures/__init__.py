"""build_model: mirror of reference pytocr/modeling/architectures/__init__.py:9-19."""
import copy

__all__ = ["build_model"]


def build_model(config):
    from .base_model import BaseModel
    config = copy.deepcopy(config)
    if "name" not in config:
        return BaseModel(config)
    name = config.pop("name")
    if name != "BaseModel":
        raise NotImplementedError("pytorchocr_amd: architecture %r is outside the accelerated hot path "
                                  "(DistillationModel is training-time only)" % name)
    return BaseModel(config)
