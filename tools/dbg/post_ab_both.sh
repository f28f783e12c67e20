#!/bin/bash
# GPU box: dbpost.hip build variants by the device time of the call on BOTH inputs: the text-like stress maps (post_device_ms.py) and the
# scene checkpoint's own maps (scene_post.py).  usage: post_ab_both.sh "<flags 1>" "<flags 2>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for FLAGS in "$@"; do
  echo "== variant: '${FLAGS}'"
  export PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS"
  touch pytorchocr_amd/csrc/dbpost.hip
  python -m pytorchocr_amd.build > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  timeout -k 10 600 python -m pytest tests/test_gpu_dbpost.py -m gpu -x -q 2>&1 | tail -1 || exit 1
  timeout -k 10 200 python3 tools/dbg/post_device_ms.py 2>&1 | grep "stress maps" || exit 1
  timeout -k 10 300 python3 tools/dbg/scene_post.py r18 2>&1 | tail -2
done
