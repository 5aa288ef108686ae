#!/bin/bash
# one PMC pass (LDS conflicts) of the step kernel for a task
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for spec in "8192 a1" "8192 anymal_c_rough" "4096 pointfoot_flat"; do
  set -- $spec
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/q_$2 -- python3 $R/tools/step_bench.py $1 30 $2 > $O/q_$2.log 2>&1
  f=$(find $O/q_$2 -name "*counter_collection.csv" | head -1)
  echo "== $2"; [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f pf_step_kernel | tail -5
  rm -rf $O/q_$2
done
