cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  tag=$(echo $set | cut -d' ' -f1)
  for shape in "32 256 184 320 64" "32 64 184 320 64"; do
    d=$R/gpurun_out/pmcw_${tag}_$(echo $shape | tr ' ' '_')
    timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/tools/run_conv_layer.py $shape > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
    python3 $R/tools/pmc_analyze.py $d | grep -A1 wino
  done
done
