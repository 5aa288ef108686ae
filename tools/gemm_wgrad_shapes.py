"""The update's three weight-gradient shapes on the bf16x6 transposed-image kernel, one network each, tiles and split counts as the
PPO path picks them (and neighbours): A/B material for changes to pf_gemm3_tn_body."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as gb
for (M, N, K, cases) in [(512, 148, 24576, ((1, 32), (1, 64))), (512, 27, 24576, ((1, 32), (1, 64))), (128, 256, 24576, ((3, 32), (3, 64), (1, 64))),
                         (256, 512, 24576, ((0, 32), (0, 64), (1, 32)))]:
    line = f"BWD_W M={M} N={N} K={K}: "
    for tile, S in cases:
        us, tf, err = gb.run(2, M, N, K, tile, splits=S, check=True, iters=40, fam=1)
        line += f"tile {tile} S={S}: {us:6.1f} us (err {err:.1e})   "
    print(line, flush=True)
