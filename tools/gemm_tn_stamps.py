"""Per-phase time of the weight-gradient kernel's slice loop (pf_gemm3_tn_body), from a diagnostic build:
   hipcc -DG3_TN_STAMPS ... -> tools/_build/libpf_tnstamps.so ;  PF_LIB_VARIANT=tools/_build/libpf_tnstamps.so python tools/gemm_tn_stamps.py
Lane 0 of wavefront 0 of every workgroup accumulates s_memtime deltas; the sums over all workgroups are read back after each case."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pointfoot_amd import _abi
_abi.lib_path = lambda: os.path.abspath(os.environ["PF_LIB_VARIANT"])
import torch
import gemm_bench as gb
lib = gb.lib
lib.pf_debug_gemm_tn_stamps.argtypes = [C.c_void_p]
names = ["barrier A", "stage (load wait + split + plane stores)", "barrier B", "issue next loads", "fragment reads + MFMAs", "prologue (first loads issued)",
         "column sums + write-out", "slices"]
buf = (C.c_ulonglong * 8)()
for (M, N, K, tile, S) in [(256, 512, 24576, 0, 32), (256, 512, 24576, 0, 64), (256, 512, 24576, 1, 32), (128, 256, 24576, 3, 64), (512, 148, 24576, 1, 32), (512, 148, 24576, 1, 64)]:
    gb.run(2, M, N, K, tile, splits=S, check=False, iters=3, fam=1)
    torch.cuda.synchronize()
    assert lib.pf_debug_gemm_tn_stamps(buf) == 0
    us, tf, err = gb.run(2, M, N, K, tile, splits=S, check=False, iters=20, fam=1)
    torch.cuda.synchronize()
    assert lib.pf_debug_gemm_tn_stamps(buf) == 0
    v = [float(x) for x in buf]
    tot, nsl = sum(v[:7]), max(v[7], 1.0)
    print(f"dW {M} x {N}, K = {K}, tile {tile}, S = {S}: {us:.1f} us per call (with the slab reduction); {nsl / 23:.0f} workgroup-slices per launch")
    for i in range(7):
        print(f"    {names[i]:45s} {100 * v[i] / tot:5.1f} %   {v[i] / nsl:7.1f} ticks per slice")
    print(f"    {'whole workgroup life':45s}          {tot / nsl:7.1f} ticks per slice", flush=True)
