"""build_post_process: mirror of reference pytocr/postprocess/__init__.py:13-30 (Global keys are merged into the kwargs)."""
import copy

__all__ = ["build_post_process"]


def build_post_process(config, global_config=None):
    from .db_postprocess import DBPostProcess
    from .cls_postprocess import ClsPostProcess
    from .rec_postprocess import CTCLabelDecode
    support = {"DBPostProcess": DBPostProcess, "CTCLabelDecode": CTCLabelDecode, "ClsPostProcess": ClsPostProcess}
    config = copy.deepcopy(config)
    name = config.pop("name")
    if global_config is not None:
        config.update(global_config)
    assert name in support, "post process only support {} (pytorchocr_amd hot path)".format(list(support))
    return support[name](**config)
