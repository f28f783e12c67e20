# (the library carries the digest of its compile flags: the flags stay exported for the runs, and the default build is restored on exit)
trap 'unset PTOCR_EXTRA_HIPCC_FLAGS; python -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for c in 1 2 4; do
  export PTOCR_EXTRA_HIPCC_FLAGS=-DPT_SEL_CHUNKS=$c; python -m pytorchocr_amd.build > gpurun_out/sel_build_$c.log 2>&1 || { tail -3 gpurun_out/sel_build_$c.log; exit 1; }
  echo "SEL_CHUNKS $c"; bash tools/dbg/post_kstats.sh 2>&1 | grep -E "select|post-process"
done
