# (the library carries the digest of its compile flags: the flags stay exported for the runs, and the default build is restored on exit)
trap 'unset PTOCR_EXTRA_HIPCC_FLAGS; python -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for r in 16 8; do
  export PTOCR_EXTRA_HIPCC_FLAGS=-DPT_SLAB_ROWS=$r; python -m pytorchocr_amd.build > gpurun_out/slab_build_$r.log 2>&1 || { tail -3 gpurun_out/slab_build_$r.log; exit 1; }
  echo "SLAB_ROWS $r"; bash tools/dbg/post_kstats.sh 2>&1 | grep -E "ccl_slab|merge_kernel<1>|flatten|post-process"
  python tools/bench_post.py 30 2>&1 | tail -1
  python -m pytest tests/test_gpu_dbpost.py -x -q 2>&1 | tail -1
done
