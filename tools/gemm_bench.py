"""GEMM micro-benchmark + correctness check of the fp32 MFMA kernel family (pf_gemm.h) against torch.
   python tools/gemm_bench.py            # PPO shapes + 4096^3, all tile variants"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointfoot_amd import _abi
lib = _abi.load()
lib.pf_gemm_test.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
dev = "cuda:0"
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(mode, M, N, K, tile, splits=128, iters=20, check=True):
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
    if mode == 0: A, B = R(M, K), R(N, K)
    elif mode == 1: A, B = R(M, K), R(K, N)
    else: A, B = R(K, M), R(K, N)
    bias, H = R(N), R(M, N)
    Cm = torch.zeros(M * N + M, device=dev)
    ws = torch.zeros(splits * (M * N + M), device=dev) if mode == 2 else torch.zeros(1, device=dev)
    call = lambda: lib.pf_gemm_test(mode, M, N, K, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), bias.data_ptr(), H.data_ptr(), ws.data_ptr(), splits, tile, stream)
    assert call() == 0, lib.pf_last_error()
    torch.cuda.synchronize()
    err = float("nan")
    if check:
        if mode == 0: ref = torch.nn.functional.elu(A @ B.t() + bias)
        elif mode == 1: ref = (A @ B) * torch.where(H > 0, torch.ones_like(H), H + 1)
        else: ref = A.t() @ B
        err = ((Cm[:M * N].view(M, N) - ref).abs().max() / ref.abs().max()).item()
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
        print(line, flush=True)

if __name__ == '__main__':
    main()
