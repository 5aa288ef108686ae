"""A/B of the two bf16x6 NT kernels on the PPO's forward / data-gradient shapes, interleaved rounds in ONE process
(cdna_hip_programming.md 5.4 rule 24): pf_gemm3_body with pre-split weight planes (pf_gemm_test mode 20/21, register staging)
against pf_gemm3f_body (mode 30/31, LDS-DMA staging), both checked against float64.
   python tools/gemm_dma_bench.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_bench as gb

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    shapes = [(0, 24576, 512, 28), (0, 24576, 512, 148), (0, 24576, 256, 512), (0, 24576, 128, 256), (1, 24576, 512, 256), (1, 24576, 256, 128),
              (0, 4096, 512, 148), (0, 4096, 256, 512), (0, 4096, 128, 256), (0, 4096, 4096, 4096),
              (0, 1000, 132, 100), (1, 777, 260, 36), (0, 130, 64, 8)]
    for mode, M, N, K in shapes:
        variants = [("reg 128x64", 2, 1), ("reg 64x64", 2, 3), ("dma 128/32", 3, 0), ("dma 64/32", 3, 1), ("dma auto", 3, -1)]
        errs, times = {}, {v[0]: [] for v in variants}
        for name, fam, tile in variants:
            _, _, errs[name] = gb.run(mode, M, N, K, tile, iters=2, check=True, fam=fam)
        for _ in range(rounds):
            for name, fam, tile in variants:
                us, _, _ = gb.run(mode, M, N, K, tile, iters=20, check=False, fam=fam)
                times[name].append(us)
        line = f"{gb.names[mode]:18s} M={M:6d} N={N:5d} K={K:5d}: "
        for name, _, _ in variants:
            ts = sorted(times[name]); med = ts[len(ts) // 2]
            line += f"{name} {med:7.1f}us (min {ts[0]:6.1f}, {2.0 * M * N * K / med / 1e6:6.1f}TF, err {errs[name]:.1e})  "
        print(line, flush=True)

if __name__ == '__main__':
    main()
