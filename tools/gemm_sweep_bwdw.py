"""Split-K sweep of the weight-gradient GEMM (64x64 tiles) over the PPO layer shapes; prints time per split count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as gb
for (M, N, K) in [(256, 512, 24576), (128, 256, 24576), (512, 148, 24576), (512, 27, 24576), (512, 235, 49152), (128, 64, 24576)]:
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    line = f"BWD_W M={M} N={N} K={K} tiles={tiles}: "
    for S in (16, 24, 32, 48, 64, 96, 128, 192, 256):
        us, tf, err = gb.run(2, M, N, K, 2, splits=S, check=False, iters=20)
        line += f"S={S}({tiles * S}):{us:6.1f} "
    print(line, flush=True)
