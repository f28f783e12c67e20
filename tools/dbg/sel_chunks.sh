for c in 1 2 4; do
  PTOCR_EXTRA_HIPCC_FLAGS=-DPT_SEL_CHUNKS=$c python -m pytorchocr_amd.build > gpurun_out/sel_build_$c.log 2>&1 || { tail -3 gpurun_out/sel_build_$c.log; exit 1; }
  echo "SEL_CHUNKS $c"; bash tools/dbg/post_kstats.sh 2>&1 | grep -E "select|post-process"
done
