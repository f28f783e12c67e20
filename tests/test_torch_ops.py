"""torch.ops.pytocr_mi355.* (SURVEY.md 8b): registration on the CPU, results against the class API and the oracle on the GPU."""
import numpy as np
import pytest
import torch


def test_ops_are_registered_and_refuse_cpu_tensors():
    import pytorchocr_amd.torch_ops  # noqa: F401
    assert hasattr(torch.ops.pytocr_mi355, "db_postprocess") and hasattr(torch.ops.pytocr_mi355, "ctc_greedy")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.pytocr_mi355.ctc_greedy(torch.zeros(3, 2, 5))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.pytocr_mi355.db_postprocess(torch.zeros(1, 1, 8, 8), torch.zeros(1, 4, dtype=torch.float64), 0.3, 0.5, 1.7, 1000, False)


@pytest.mark.gpu
def test_custom_ops_match_the_class_api_and_the_oracle():
    import pytorchocr_amd.torch_ops  # noqa: F401
    from oracle import ctc_oracle, dbpost
    from pytorchocr_amd.postprocess import build_post_process
    from pytorchocr_amd.utils.synth import synth_prob_maps, uniform
    pm = synth_prob_maps(3, 96, 160, seed=4)
    maps = torch.from_numpy(pm[:, None]).cuda()
    shape_list = torch.tensor([[96, 160, 1.0, 1.0], [192, 320, 2.0, 2.0], [48, 80, 0.5, 0.5]], dtype=torch.float64)
    boxes, counts = torch.ops.pytocr_mi355.db_postprocess(maps, shape_list, 0.3, 0.5, 1.7, 1000, False)
    assert boxes.dtype == torch.int16 and counts.dtype == torch.int32 and boxes.shape == (int(counts.sum()), 4, 2)
    post = build_post_process(dict(name="DBPostProcess", thresh=0.3, box_thresh=0.5, unclip_ratio=1.7, cpp_speedup=True), {})
    res = post({"maps": maps}, shape_list.numpy())
    off = 0
    for i in range(3):
        k = int(counts[i])
        exp = dbpost.boxes_from_bitmap(pm[i], dbpost.binarize(pm[i], 0.3), 0.5, 1.7, int(shape_list[i, 1]), int(shape_list[i, 0]))
        assert np.array_equal(boxes[off:off + k].numpy(), res[i]["points"]) and np.array_equal(boxes[off:off + k].numpy().astype(np.int32), exp)
        off += k
    assert off > 10
    with pytest.raises(RuntimeError, match="max_candidates"):
        torch.ops.pytocr_mi355.db_postprocess(maps, shape_list, 0.3, 0.5, 1.7, 500, False)
    # ctc_greedy on probabilities [T,B,C]: arg-max / max over C per (b,t)
    T, B, Cn = 9, 4, 37
    p = uniform((T, B, Cn), 3, 0.0, 1.0)
    p = (p / p.sum(2, keepdims=True)).astype(np.float32)
    idx, prob = torch.ops.pytocr_mi355.ctc_greedy(torch.from_numpy(p).cuda())
    assert idx.shape == (B, T) and idx.dtype == torch.int32
    assert np.array_equal(idx.cpu().numpy(), p.transpose(1, 0, 2).argmax(2)) and np.array_equal(prob.cpu().numpy(), p.transpose(1, 0, 2).max(2))
    texts = ctc_oracle.decode(idx.cpu().numpy(), prob.cpu().numpy(), ["blank"] + list("0123456789abcdefghijklmnopqrstuvwxyz")) if hasattr(ctc_oracle, "decode") else None
    assert texts is None or len(texts) == B
