#!/bin/bash
# PMC passes on the env-step kernel: bash tools/pmc_step.sh <tag> [N] [task]
TAG=$1; N=${2:-4096}; TASK=${3:-pointfoot_rough}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc$i -- python3 $R/tools/step_bench.py $N 40 $TASK > $O/${TAG}_pmc$i.log 2>&1
  f=$(find $O/${TAG}_pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f pf_step_kernel | tail -8
  rm -rf $O/${TAG}_pmc$i
done
