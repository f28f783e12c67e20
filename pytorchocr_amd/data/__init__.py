"""create_operators / transform: mirror of reference pytocr/data/imaug/__init__.py:19-48 for the inference operators
(Global keys are merged into every operator's kwargs, hence the **kwargs everywhere)."""
from .imaug import ClsResizeImg, DecodeImage, DetResizeForTest, KeepKeys, Normalize, RecResizeImg, RecResizeImgForTest, ToTensor  # noqa: F401
from .label_ops import ClsLabelEncode, CTCLabelEncode, DetLabelEncode  # noqa: F401

_OPS = {"DecodeImage": DecodeImage, "DetResizeForTest": DetResizeForTest, "ToTensor": ToTensor, "Normalize": Normalize,
        "KeepKeys": KeepKeys, "RecResizeImg": RecResizeImg, "ClsResizeImg": ClsResizeImg, "RecResizeImgForTest": RecResizeImgForTest,
        "DetLabelEncode": DetLabelEncode, "CTCLabelEncode": CTCLabelEncode, "ClsLabelEncode": ClsLabelEncode}


def transform(data, ops=None):
    if ops is None:
        ops = []
    for op in ops:
        data = op(data)
        if data is None:
            return None
    return data


def create_operators(op_param_list, global_config=None):
    assert isinstance(op_param_list, list), "operator config should be a list"
    ops = []
    for operator in op_param_list:
        assert isinstance(operator, dict) and len(operator) == 1, "yaml format error"
        op_name = list(operator)[0]
        param = {} if operator[op_name] is None else dict(operator[op_name])
        if global_config is not None:
            param.update(global_config)
        if op_name not in _OPS:
            raise NotImplementedError("pytorchocr_amd: operator %r is outside the inference hot path" % op_name)
        ops.append(_OPS[op_name](**param))
    return ops
