#!/bin/bash
# Build a second copy of the library with extra flags (A/B runs of a kernel variant on ONE box in one gpurun call):
#   tools/build_variant.sh gpurun_out/libpf_variant.so -DPF_POINT_DUAL=1
# SCHED=<strategy|none> overrides the env-step translation unit's -amdgpu-sched-strategy (default iterative-ilp).
# Tools that take PF_LIB_VARIANT=<path> (tools/step_bench.py, tests via conftest) load it instead of the product library.
set -e
cd "$(dirname "$0")/.."
out=$1; shift
mkdir -p "$(dirname "$out")" tools/_build
sched="-mllvm -amdgpu-sched-strategy=${SCHED:-iterative-ilp}"
[ "${SCHED}" = none ] && sched=""
obj=tools/_build/$(basename "$out" .so).o
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -mllvm -inline-threshold=225 -fPIC $sched "$@" -c -o "$obj" pointfoot_amd/csrc/pf_env.hip
[ -f pointfoot_amd/csrc/_obj/pf_ppo.hip.o ] || python -c "import __graft_entry__ as g; g.build_hip()"
hipcc --offload-arch=gfx950 -fPIC -shared -o "$out" "$obj" pointfoot_amd/csrc/_obj/pf_ppo.hip.o
echo "built $out"
