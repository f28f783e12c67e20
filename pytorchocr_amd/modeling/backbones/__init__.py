"""build_backbone: mirror of reference pytocr/modeling/backbones/__init__.py:3-28 for the hot-path backbones."""
__all__ = ["build_backbone"]


def build_backbone(config, model_type):
    from .det_mobilenet_v3 import MobileNetV3
    from .det_resnet import ResNet
    from .rec_mobilenet_v3 import MobileNetV3 as RecMobileNetV3
    from .rec_vgg import VGG
    support = {"det": {"ResNet": ResNet, "MobileNetV3": MobileNetV3}, "rec": {"VGG": VGG}, "cls": {"MobileNetV3": RecMobileNetV3}}
    if model_type not in support:
        raise NotImplementedError("pytorchocr_amd: model_type %r is outside the accelerated hot path" % model_type)
    config = dict(config)
    name = config.pop("name")
    assert name in support[model_type], "when model_type is {}, backbone only support {} (pytorchocr_amd hot path)".format(
        model_type, list(support[model_type]))
    return support[model_type][name](**config)
