# per-kernel times of the stand-alone post-process under PTOCR_DBPOST_DBG_SKIP knock-outs (16: no rectangle of the offset polygon, 32: no unclip at all)
cd /tmp && export TMPDIR=/tmp
for d in 0 64 128; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pk$d
  PTOCR_DBPOST_DBG_SKIP=$d rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pk$d -- python3 $GRAFT_REPO_ROOT/tools/bench_post.py 10 > $GRAFT_REPO_ROOT/gpurun_out/pk$d.log 2>&1
  echo "DBG_SKIP=$d"; grep "unclip_kernel\|rect_kernel\|hull_kernel\|score_kernel" $(ls $GRAFT_REPO_ROOT/gpurun_out/pk$d/*/*kernel_stats.csv | head -1) | awk -F, '{print $1, $4}' | cut -c1-100
done
