#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace summaries of the bench workloads + HBM-traffic PMC passes (separate passes,
# --kernel-trace only: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Outputs land in gpurun_out/profiles_new/; copy what should
# be judged into profiles/.   usage: collect_profiles.sh [round tag, default r02]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
O=$R/gpurun_out/profiles_new
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
kstats() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/${TAG}_${name}_stdout.log 2>&1
  cp $(ls $O/$name/*/*kernel_stats.csv | head -1) $O/${TAG}_${name}_kernel_stats.csv
  rm -rf $O/$name
}
kstats bench_det_b32 python3 $R/bench.py --steps 10 --warmup 3 --cpu-images 0 --cpu-lines 0 --crnn-steps 0 --no-embed
kstats bench_crnn_b512 python3 $R/bench.py --workload crnn --steps 10 --warmup 3 --cpu-lines 0
kstats bench_mbv3s_bf16_b32 python3 $R/bench.py --det-model mbv3s --dtype bf16 --steps 10 --warmup 3 --cpu-images 0
kstats bench_ocr_64 python3 $R/bench.py --workload ocr --steps 3 --warmup 1 --cpu-images 0
kstats post_standalone python3 $R/tools/bench_post.py 20
pmc() {  # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${name}_$ctr -- "$@" > $O/pmc_${name}_$ctr.log 2>&1
}
for c in FETCH_SIZE WRITE_SIZE; do
  pmc det $c python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --cpu-lines 0 --crnn-steps 0 --no-embed
  pmc crnn $c python3 $R/bench.py --workload crnn --steps 2 --warmup 1 --cpu-lines 0
  pmc post $c python3 $R/tools/bench_post.py 3
  pmc bf16 $c python3 $R/bench.py --det-model mbv3s --dtype bf16 --steps 2 --warmup 1 --post-input none --cpu-images 0
done
python3 $R/tools/traffic_from_pmc.py $O/pmc_det_FETCH_SIZE $O/pmc_det_WRITE_SIZE conv_wino4r_kernel $O/conv_traffic.json
python3 $R/tools/traffic_from_pmc.py $O/pmc_crnn_FETCH_SIZE $O/pmc_crnn_WRITE_SIZE conv_wino4r_kernel $O/crnn_traffic.json
python3 $R/tools/pmc_analyze.py $O/pmc_det_FETCH_SIZE > $O/${TAG}_pmc_det_fetch_size.txt
python3 $R/tools/pmc_analyze.py $O/pmc_det_WRITE_SIZE > $O/${TAG}_pmc_det_write_size.txt
python3 $R/tools/pmc_analyze.py $O/pmc_post_FETCH_SIZE > $O/${TAG}_pmc_post_fetch_size.txt
python3 $R/tools/pmc_analyze.py $O/pmc_post_WRITE_SIZE > $O/${TAG}_pmc_post_write_size.txt
python3 $R/tools/pmc_analyze.py $O/pmc_bf16_FETCH_SIZE > $O/${TAG}_pmc_bf16_fetch_size.txt
python3 $R/tools/pmc_analyze.py $O/pmc_bf16_WRITE_SIZE > $O/${TAG}_pmc_bf16_write_size.txt
python3 $R/tools/post_hbm_from_pmc.py $O/${TAG}_pmc_post_fetch_size.txt $O/${TAG}_pmc_post_write_size.txt $O/post_traffic.json > $O/${TAG}_post_hbm_gbps.txt
python3 $R/tools/post_hbm_from_pmc.py $O/${TAG}_pmc_bf16_fetch_size.txt $O/${TAG}_pmc_bf16_write_size.txt $O/mbv3s_bf16_traffic.json bf16 > $O/${TAG}_bf16_hbm_gbps.txt
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
ls $O
