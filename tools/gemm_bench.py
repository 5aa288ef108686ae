"""GEMM micro-benchmark + correctness check of both kernel families -- fp32 MFMA (pf_gemm.h) and fp32 by exact bf16 split on
the bf16 MFMA (pf_gemm3.h, "bf16x6") -- against a float64 reference.
   python tools/gemm_bench.py            # PPO shapes + 4096^3, all tile variants, both families"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointfoot_amd import _abi
lib = _abi.load()
lib.pf_gemm_test.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
dev = "cuda:0"
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(mode, M, N, K, tile, splits=128, iters=20, check=True, fam=0):
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
    if mode == 0: A, B = R(M, K), R(N, K)
    elif mode == 1: A, B = R(M, K), R(K, N)
    else: A, B = R(K, M), R(K, N)
    bias, H = R(N), R(M, N)
    Cm = torch.zeros(M * N + M, device=dev)
    ws = torch.zeros(splits * (M * N + M), device=dev) if mode == 2 else torch.zeros(2 * N * (K + 8) + 64, device=dev)
    call = lambda: lib.pf_gemm_test(mode + 10 * fam, M, N, K, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), bias.data_ptr(), H.data_ptr(), ws.data_ptr(), splits, tile, stream)
    assert call() == 0, lib.pf_last_error()
    torch.cuda.synchronize()
    err = float("nan")
    if check:
        A6, B6 = A.double(), B.double()
        if mode == 0: ref = torch.nn.functional.elu(A6 @ B6.t() + bias.double())
        elif mode == 1: ref = (A6 @ B6) * torch.where(H > 0, torch.ones_like(H), H + 1).double()
        else: ref = A6.t() @ B6
        err = ((Cm[:M * N].view(M, N).double() - ref).abs().max() / ref.abs().max()).item()
        if mode == 2:
            err = max(err, ((Cm[M * N:] - A.sum(0)).abs().max() / A.sum(0).abs().max()).item())
    for _ in range(3): call()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    return us, 2.0 * M * N * K / us / 1e6, err

names = {0: "FWD  x W^T+b,ELU", 1: "BWD_X g W *ELU'", 2: "BWD_W g^T x (+colsum, splitK)"}
def main():
    shapes = [(0, 4096, 4096, 4096), (0, 24576, 512, 148), (0, 24576, 256, 512), (0, 24576, 128, 256), (0, 24576, 512, 27), (0, 4096, 512, 148), (0, 4096, 256, 512),
              (1, 24576, 512, 256), (1, 24576, 256, 128), (2, 256, 512, 24576), (2, 128, 256, 24576), (2, 512, 148, 24576), (2, 512, 27, 24576)]
    for mode, M, N, K in shapes:
        line = f"{names[mode]:32s} M={M:6d} N={N:5d} K={K:6d}: "
        for tile, tn in ((0, "128x128"), (1, "128x64"), (2, "64x64"), (-1, "auto")):
            us, tf, err = run(mode, M, N, K, tile, check=(tile in (0, -1)))
            line += f"{tn} {us:7.1f}us {tf:6.1f}TF " + (f"(err {err:.1e}) " if err == err else "")
        print("fp32   " + line, flush=True)
        line = f"{names[mode]:32s} M={M:6d} N={N:5d} K={K:6d}: "
        for tile, tn in ((0, "128x128"), (1, "128x64"), (2, "64x128"), (3, "64x64")) + ((-1, "auto"),):
            us, tf, err = run(mode, M, N, K, tile, check=(tile in (0, -1)), fam=1)
            line += f"{tn} {us:7.1f}us {tf:6.1f}TF " + (f"(err {err:.1e}) " if err == err else "")
        print("bf16x6 " + line, flush=True)
        if mode < 2:
            line = f"{names[mode]:32s} M={M:6d} N={N:5d} K={K:6d}: "
            for tile, tn in ((0, "128x128"), (1, "128x64"), (2, "64x128"), (3, "64x64"), (-1, "auto")):
                us, tf, err = run(mode, M, N, K, tile, check=(tile in (0, -1)), fam=2)
                line += f"{tn} {us:7.1f}us {tf:6.1f}TF " + (f"(err {err:.1e}) " if err == err else "")
            print("x6+Wpl " + line, flush=True)

if __name__ == '__main__':
    main()
