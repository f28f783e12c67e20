"""build_neck: mirror of reference pytocr/modeling/necks/__init__.py:3-13 for the hot-path necks."""
__all__ = ["build_neck"]


def build_neck(config):
    from .fpn import FPN
    from .rnn import SequenceEncoder
    support = {"FPN": FPN, "SequenceEncoder": SequenceEncoder}
    config = dict(config)
    name = config.pop("name")
    assert name in support, "neck only support {} (pytorchocr_amd hot path)".format(list(support))
    return support[name](**config)
