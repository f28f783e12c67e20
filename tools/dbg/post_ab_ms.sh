#!/bin/bash
# GPU box: A/B of dbpost.hip compile-time variants by the DEVICE time of the stand-alone call (no profiler).
# usage: post_ab_ms.sh "<flags of variant 1>" "<flags of variant 2>" ...   ("" = the default build; the library left behind is the LAST variant's:
# rebuild with `python -m pytorchocr_amd.build` afterwards -- the snapshot on the box is thrown away anyway)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for FLAGS in "$@"; do
  echo "== variant: '${FLAGS}'"
  export PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS"
  touch pytorchocr_amd/csrc/dbpost.hip
  python -m pytorchocr_amd.build > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  timeout -k 10 600 python -m pytest tests/test_gpu_dbpost.py -m gpu -x -q 2>&1 | tail -1 || exit 1
  for i in 1 2; do timeout -k 10 200 python3 tools/dbg/post_device_ms.py 2>&1 | grep "stress maps" || exit 1; done
done
