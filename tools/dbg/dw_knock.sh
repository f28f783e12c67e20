# (the library carries the digest of its compile flags: the flags stay exported for the runs, and the default build is restored on exit)
trap 'unset PTOCR_EXTRA_HIPCC_FLAGS; python -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
# depthwise knock-outs: bash tools/dbg/dw_knock.sh  (rebuilds the library on the box with -DDW_DBG=n, prints the dw kernel times)
for d in 0 1 2 4 3; do
  touch pytorchocr_amd/csrc/bf16_ops.hip; export PTOCR_EXTRA_HIPCC_FLAGS="-DDW_DBG=$d"; python -m pytorchocr_amd.build > /dev/null 2>&1 || exit 1
  echo "DW_DBG=$d"
  bash tools/dbg/bf16_trace.sh | grep "dwconv\|sum"
done
