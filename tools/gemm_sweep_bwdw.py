import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as gb
for (M, N, K) in [(256, 512, 24576), (128, 256, 24576), (512, 148, 24576), (512, 27, 24576), (128, 64, 24576), (64, 32, 24576), (128, 27, 24576)]:
    best = None
    for tile, tn in ((0, "128x128"), (1, "128x64"), (2, "64x64")):
        line = f"BWD_W M={M} N={N} {tn}: "
        for S in (4, 8, 16, 32, 64):
            us, tf, err = gb.run(2, M, N, K, tile, splits=S, check=False, iters=10)
            line += f"S={S}:{us:6.1f} "
            if best is None or us < best[0]: best = (us, tn, S)
        print(line, flush=True)
    print("   best", best, flush=True)
