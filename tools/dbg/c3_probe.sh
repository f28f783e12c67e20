# (the library carries the digest of its compile flags: the flags stay exported for the runs, and the default build is restored on exit)
trap 'unset PTOCR_EXTRA_HIPCC_FLAGS; python -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
touch pytorchocr_amd/csrc/bf16_ops.hip; export PTOCR_EXTRA_HIPCC_FLAGS="-DC3_DBG=16"; python -m pytorchocr_amd.build > /dev/null 2>&1 || exit 1
python tools/dbg/c3_probe.py
