#!/bin/bash
# Build the ablated variants of the LDS-DMA bf16-split GEMM (tools/gemm_dma_ablate.hip) -- "build" here (hipcc cross-compiles),
# "run" on the GPU box: one time per removed ingredient for the PPO's dominant shapes.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
MASKS="${MASKS:-0 1 2 4 8 16 24 32 30 62 63}"
if [ "$1" = build ]; then
  for m in $MASKS; do
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DG3F_ABLATE=$m -o tools/_build/gemm_dma_ablate_$m tools/gemm_dma_ablate.hip &
    if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
  done
  wait
else
  for shape in "0 24576 256 512 0 0 1" "0 24576 256 512 1 0 1" "1 24576 512 256 0 0 1" "0 24576 512 148 0 0 1" "0 24576 128 256 1 0 1" "1 24576 256 128 0 0 1" "1 24576 256 128 1 0 1" ${BIG:+"0 4096 4096 4096 0"}; do
    for m in $MASKS; do tools/_build/gemm_dma_ablate_$m $shape; done
  done
fi
