"""Evaluation loop with the contract of the reference's `eval` (tools/program.py:421-473): iterate a loader of LIST batches
([images, shape_list / labels, ...], tensors or arrays), run the model in eval mode without gradients, post-process, feed the
metric, return its totals plus `fps` (frames / seconds spent in model + host copy, as the reference times it).

The model and post-process are this package's (HIP path); the loader is anything iterable with __len__ (a torch DataLoader over
the reference's SimpleDataSet, or a plain list of batches as in tests/)."""
import time

import numpy as np
import torch


def _to_numpy(item):
    if isinstance(item, torch.Tensor):
        return item.detach().cpu().numpy()
    return item.numpy() if hasattr(item, "numpy") else item


def eval(model, device, valid_dataloader, post_process_class, eval_class, model_type=None, extra_input=False):  # noqa: A001
    if model_type in ("table", "kie") or extra_input:
        raise NotImplementedError("pytorchocr_amd eval: table / kie / extra-input models are outside the hot path")
    was_training = model.training
    model.eval()
    frames, seconds = 0.0, 0.0
    with torch.no_grad():
        for batch in valid_dataloader:
            batch = [item.to(device) if isinstance(item, torch.Tensor) else item for item in batch]
            images = batch[0]
            t0 = time.time()
            preds = model(images)
            batch = [_to_numpy(item) for item in batch]
            seconds += time.time() - t0
            eval_class(post_process_class(preds, batch[1]), batch)
            frames += len(images)
        metric = eval_class.get_metric()
    if was_training:
        model.train()
    metric["fps"] = frames / seconds if seconds > 0 else float("nan")
    return metric
