#!/bin/bash
# Build a second copy of the library with extra flags (A/B runs of a kernel variant on ONE box in one gpurun call):
#   tools/build_variant.sh gpurun_out/libpf_variant.so -DPF_POINT_DUAL=1
# SCHED=<strategy|none> overrides the quadruped translation unit's -amdgpu-sched-strategy (default iterative-ilp);
# SCHED_POINT=<strategy> sets one for pf_env.hip (the PointFoot instance; default: the compiler's).
# Tools that take PF_LIB_VARIANT=<path> (tools/step_bench.py, tests via conftest) load it instead of the product library.
set -e
cd "$(dirname "$0")/.."
out=$1; shift
mkdir -p "$(dirname "$out")" tools/_build
sched="-mllvm -amdgpu-sched-strategy=${SCHED:-iterative-ilp}"      # the quadruped translation unit's scheduler
[ "${SCHED}" = none ] && sched=""
base=tools/_build/$(basename "$out" .so)
F="-O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -mllvm -inline-threshold=225 -fPIC"
hipcc $F ${SCHED_POINT:+-mllvm -amdgpu-sched-strategy=$SCHED_POINT} "$@" -c -o "$base.env.o" pointfoot_amd/csrc/pf_env.hip &
hipcc $F $sched "$@" -c -o "$base.quad.o" pointfoot_amd/csrc/pf_env_quad.hip &
wait
[ -f pointfoot_amd/csrc/_obj/pf_ppo.hip.o ] || python -c "import __graft_entry__ as g; g.build_hip()"
hipcc --offload-arch=gfx950 -fPIC -shared -o "$out" "$base.env.o" "$base.quad.o" pointfoot_amd/csrc/_obj/pf_ppo.hip.o
echo "built $out"
