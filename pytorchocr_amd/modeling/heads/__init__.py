"""build_head: mirror of reference pytocr/modeling/heads/__init__.py:3-26 for the hot-path heads."""
__all__ = ["build_head"]


def build_head(config):
    from .cls_head import ClsHead
    from .det_db_head import DBHead
    from .rec_ctc_head import CTCHead
    support = {"DBHead": DBHead, "CTCHead": CTCHead, "ClsHead": ClsHead}
    config = dict(config)
    name = config.pop("name")
    assert name in support, "head only support {} (pytorchocr_amd hot path)".format(list(support))
    return support[name](**config)
