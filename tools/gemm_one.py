"""Run one GEMM shape of the pf_gemm family repeatedly (for rocprofv3 --pmc passes).
   python tools/gemm_one.py MODE M N K [TILE] [ITERS] [FAM]      MODE 0 FWD, 1 BWD_X, 2 BWD_W; FAM 0 fp32 MFMA, 1 bf16x6"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as gb
mode, M, N, K = (int(x) for x in sys.argv[1:5])
tile = int(sys.argv[5]) if len(sys.argv) > 5 else -1
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 50
fam = int(sys.argv[7]) if len(sys.argv) > 7 else 0
us, tf, err = gb.run(mode, M, N, K, tile, iters=iters, check=False, fam=fam)
print(f"mode {mode} M={M} N={N} K={K} tile {tile}: {us:.1f} us  {tf:.1f} TFLOP/s")
