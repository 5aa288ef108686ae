#!/bin/bash
# PMC passes over a short bench run, summarised for the kernels whose name contains <pattern>:
#   bash tools/pmc_bench.sh <tag> <pattern> [bench args]
TAG=$1; PAT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_alt "$@" > $O/${TAG}_pmc$i.log 2>&1
  f=$(find $O/${TAG}_pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f "$PAT"
  rm -rf $O/${TAG}_pmc$i          # (raw per-dispatch rows: tens of MB; gpurun carries at most 64 MiB back)
done
