#!/bin/bash
# HBM traffic of the step kernel for one task (two separate --pmc passes, guide's gfx950 correction in tools/pmc_traffic.py)
#   bash tools/pmc_traffic_task.sh <task> <num_envs> <kernel-prefix> <out.json>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
task=$1; n=$2; pre=$3; out=$(realpath -m $4)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/t_fetch -- python3 $R/tools/step_bench.py $n 40 $task > $O/t_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/t_write -- python3 $R/tools/step_bench.py $n 40 $task > $O/t_write.log 2>&1
python3 $R/tools/pmc_traffic.py $O/t_fetch $O/t_write "$pre" $task $n $out
rm -rf $O/t_fetch $O/t_write
