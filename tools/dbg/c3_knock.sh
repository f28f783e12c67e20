# (the library carries the digest of its compile flags: the flags stay exported for the runs, and the default build is restored on exit)
trap 'unset PTOCR_EXTRA_HIPCC_FLAGS; python -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
# 3x3 bf16 knock-outs: rebuilds with -DC3_DBG=n on the box
for d in 0 1 2 4 8 7; do
  touch pytorchocr_amd/csrc/bf16_ops.hip; export PTOCR_EXTRA_HIPCC_FLAGS="-DC3_DBG=$d"; python -m pytorchocr_amd.build > /dev/null 2>&1 || exit 1
  echo "C3_DBG=$d"; python tools/dbg/c3_times.py
done
