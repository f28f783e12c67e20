#!/bin/bash
# GPU box: A/B of dbpost.hip compile-time variants.  usage: post_variant_ab.sh <tag> "<extra hipcc flags>"  -- rebuilds the library with the
# flags, runs the dbpost parity tests, the stand-alone kernel stats (tools/prof_post.sh) and the stage stamps; the default build is restored
# by the next `python -m pytorchocr_amd.build`
TAG=$1; FLAGS=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS" python -m pytorchocr_amd.build > gpurun_out/${TAG}_build.log 2>&1 || { tail -5 gpurun_out/${TAG}_build.log; exit 1; }
export PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS"
bash tools/prof_post.sh $TAG | grep -v "calls    [12] avg"
python tools/dbg/post_stamps.py > gpurun_out/${TAG}_stamps.log 2>&1
sed -n 2,14p gpurun_out/${TAG}_stamps.log
grep -A6 "^wave kernel" gpurun_out/${TAG}_stamps.log
