#!/bin/bash
# Build the ablated variants of the bf16-split GEMM (tools/gemm_ablate.hip) -- run with "build" here (hipcc cross-compiles),
# and with "run" on the GPU box: prints one time per removed ingredient for the PPO's dominant shapes.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
MASKS="0 1 2 4 6 8 16 32 63 62"
if [ "$1" = build ]; then
  for m in $MASKS; do
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DG3_ABLATE=$m -o tools/_build/gemm_ablate_$m tools/gemm_ablate.hip &
    if (( $(jobs -r | wc -l) >= 5 )); then wait -n; fi
  done
  wait
else
  SHAPES=${SHAPES:-"0 24576 256 512 1 0|0 24576 256 512 1 1|1 24576 512 256 1 1|0 4096 4096 4096 0 0|2 256 512 24576 0 0"}
  IFS='|' read -ra LIST <<< "$SHAPES"
  for shape in "${LIST[@]}"; do
    for m in $MASKS; do tools/_build/gemm_ablate_$m $shape; done
  done
fi
