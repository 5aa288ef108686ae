#!/bin/bash
# HBM traffic of the step kernel for one task (two separate --pmc passes, guide's gfx950 correction in tools/pmc_traffic.py)
#   bash tools/pmc_traffic_task.sh <task> <num_envs> <kernel-prefix> <out.json> [bench]
#   (5th argument `bench`: the launches of bench.py's rollout instead of tools/step_bench.py -- the env step WITH the policy's tail, pf_step_policy)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
task=$1; n=$2; pre=$3; out=$(realpath -m $4)
cd /tmp && export TMPDIR=/tmp
PROG="$R/tools/step_bench.py $n 40 $task"
[ "$5" = bench ] && PROG="$R/bench.py --task $task --num_envs $n --steps 4 --warmup 1 --no_cpu_baseline --no_alt"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/t_fetch -- python3 $PROG > $O/t_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/t_write -- python3 $PROG > $O/t_write.log 2>&1
# request sizes (tools/calib_gather.py: every line miss is one 128-byte request): the exact byte counts
rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --output-format csv -d $O/t_rq1 -- python3 $PROG > $O/t_rq1.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/t_rq2 -- python3 $PROG > $O/t_rq2.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/t_rq3 -- python3 $PROG > $O/t_rq3.log 2>&1
python3 $R/tools/pmc_traffic.py $O/t_fetch $O/t_write "$pre" $task $n $out $O/t_rq1 $O/t_rq2 $O/t_rq3
rm -rf $O/t_fetch $O/t_write $O/t_rq1 $O/t_rq2 $O/t_rq3
