#!/bin/bash
# Runs on the GPU box (gpurun): headline bench, rocprofv3 kernel trace of the same command, HBM PMC passes of the step kernel.
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -e
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
tail -c 600 $O/${TAG}_bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/tools/step_bench.py 4096 40 > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/tools/step_bench.py 4096 40 > $O/${TAG}_pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write pf_step_kernel pointfoot_rough 4096 $O/${TAG}_step_kernel_hbm_pmc.json
find $O/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
head -12 $O/${TAG}_kernel_stats.csv
