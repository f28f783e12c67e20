touch pytorchocr_amd/csrc/bf16_ops.hip; PTOCR_EXTRA_HIPCC_FLAGS="-DC3_DBG=16" python -m pytorchocr_amd.build > /dev/null 2>&1 || exit 1
python tools/dbg/c3_probe.py
