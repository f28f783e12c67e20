# SQ occupancy / issue / wait counters of the post-process kernels on the stress maps (own passes, --kernel-trace only): bash tools/dbg/pmc_sq_post.sh [tag]
# Three passes of up to six counters; prints every dbpost kernel's rows (mean per launch, summed over the chip) into gpurun_out/<tag>_pmc_sq_post.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_pmc_sq_post.txt
: > $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  O=$R/gpurun_out/pmc_sq_post_$i
  rm -rf $O
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O -- python3 $R/tools/bench_post.py 3 > $O.log 2>&1
  echo "---- pass $i: $set" >> $OUT
  python3 $R/tools/pmc_analyze.py $O >> $OUT
done
grep -A2 "border_stage\|border_states\|ccl_slab" $OUT | head -60
