#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace summaries of the two bench workloads + HBM-traffic PMC passes of the
# dominant kernel.  Outputs land in gpurun_out/profiles_new/; copy what should be judged into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_new
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/det -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-images 0 > $O/bench_det_stdout.log 2>&1
cp $(ls $O/det/*/*kernel_stats.csv | head -1) $O/bench_det_b32_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/crnn -- python3 $R/bench.py --workload crnn --steps 5 --warmup 2 --cpu-lines 0 > $O/bench_crnn_stdout.log 2>&1
cp $(ls $O/crnn/*/*kernel_stats.csv | head -1) $O/bench_crnn_b512_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 > $O/pmc_write.log 2>&1
python3 $R/tools/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write conv_wino_kernel $O/conv_traffic.json
python3 $R/tools/pmc_analyze.py $O/pmc_fetch > $O/pmc_fetch_size.txt
python3 $R/tools/pmc_analyze.py $O/pmc_write > $O/pmc_write_size.txt
rm -rf $O/det $O/crnn $O/pmc_fetch $O/pmc_write
tail -1 $O/bench_det_stdout.log
