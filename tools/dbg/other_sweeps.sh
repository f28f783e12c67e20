#!/bin/bash
# GPU box: the seeded sweeps of everything but the post-process (conv ops, cv2 boundary, models, LSTM, bf16) on the round's final code
source tools/gpu_steps.sh sweeps
export PTOCR_CONV_FUZZ=1500 PTOCR_CV2_FUZZ=2000 PTOCR_MODEL_FUZZ=150 PTOCR_LSTM_FUZZ=200 PTOCR_BF16_FUZZ=300
TAILN=3 step other_sweeps 1100 python -m pytest tests/test_gpu_conv_ops.py tests/test_gpu_entrypoints.py tests/test_gpu_det_model.py tests/test_gpu_crnn.py tests/test_gpu_bf16.py -m gpu -q -x
