#!/bin/bash
# Runs on the GPU box (gpurun): regenerates the round's evidence from the binary that is in the tree --
#   headline bench (+ fp32-MFMA alt, cpu baseline), rocprofv3 kernel trace of the same command, HBM PMC passes of both step-kernel
#   instances, instruction-mix PMC of the PointFoot step kernel, and MFMA / VALU / LDS PMC passes over the update's GEMM kernels.
# usage: bash tools/profile_round.sh <tag> [parts]     parts: any of bench trace hbm stepmix gemm anymal flat (default: all)
#   -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
set -e
TAG=${1:-r4}
PARTS=${2:-"bench trace hbm stepmix gemm anymal flat"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  python3 $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
  tail -c 400 $O/${TAG}_bench_n1.json; echo
fi
if has trace; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err
  find $O/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
  rm -rf $O/${TAG}_prof             # (the raw trace is tens of MB; gpurun carries at most 64 MiB back)
  head -14 $O/${TAG}_bench_kernel_stats.csv | cut -c1-160
fi
if has hbm; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/tools/step_bench.py 4096 40 > $O/${TAG}_pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/tools/step_bench.py 4096 40 > $O/${TAG}_pmc_write.log 2>&1
  python3 $R/tools/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write "pf_step_kernel<9" pointfoot_rough 4096 $O/${TAG}_step_kernel_hbm_pmc.json
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch_q -- python3 $R/tools/step_bench.py 8192 40 anymal_c_rough > $O/${TAG}_pmc_fetch_q.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write_q -- python3 $R/tools/step_bench.py 8192 40 anymal_c_rough > $O/${TAG}_pmc_write_q.log 2>&1
  python3 $R/tools/pmc_traffic.py $O/${TAG}_pmc_fetch_q $O/${TAG}_pmc_write_q "pf_step_kernel<17" anymal_c_rough 8192 $O/${TAG}_step_kernel_hbm_pmc_anymal_c_rough.json
  rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_fetch_q $O/${TAG}_pmc_write_q
fi
if has stepmix; then
  bash $R/tools/pmc_step.sh ${TAG}_step 4096 pointfoot_rough > $O/${TAG}_step_kernel_pmc_raw.txt 2>&1
  tail -40 $O/${TAG}_step_kernel_pmc_raw.txt
fi
if has gemm; then
  bash $R/tools/pmc_bench.sh ${TAG}_gemm gemm3 > $O/${TAG}_gemm_pmc_raw.txt 2>&1
  tail -60 $O/${TAG}_gemm_pmc_raw.txt
fi
if has anymal; then
  python3 $R/bench.py --task anymal_c_rough --num_envs 8192 --steps 10 --warmup 3 --no_cpu_baseline > $O/${TAG}_bench_anymal_c_rough_n8192.json 2> $O/${TAG}_bench_anymal.err
  tail -c 300 $O/${TAG}_bench_anymal_c_rough_n8192.json; echo
fi
if has flat; then
  python3 $R/bench.py --task pointfoot_flat --steps 10 --warmup 3 --no_cpu_baseline > $O/${TAG}_bench_pointfoot_flat.json 2> $O/${TAG}_bench_flat.err
  tail -c 300 $O/${TAG}_bench_pointfoot_flat.json; echo
fi
