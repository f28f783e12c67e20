#!/bin/bash
# gpurun helper: dbpost parity tests, then the device time of the stand-alone post-process call WITHOUT a profiler: three runs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_dbpost.py -m gpu -x -q 2>&1 | tail -2 || exit 1
for i in 1 2 3; do timeout -k 10 200 python3 tools/dbg/post_device_ms.py 2>&1 | grep "stress maps" || exit 1; done
