"""FETCH_SIZE calibration on the step kernel's read pattern: stream N records of 160 floats (dword per lane, 32 lanes per row)
from a buffer larger than the Infinity Cache, under `rocprofv3 --pmc FETCH_SIZE`.  Known bytes = rows * 640.
   rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/calib_fetch.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointfoot_amd import _abi
lib = _abi.load()
lib.pf_debug_calib_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
rows, rf = 1 << 20, 160            # 671 MB: nothing stays resident between launches
src = torch.randn(rows, rf, device="cuda:0")
sink = torch.zeros(4, device="cuda:0")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(6):
    assert lib.pf_debug_calib_read(src.data_ptr(), rows, rf, sink.data_ptr(), s) == 0
torch.cuda.synchronize()
print(f"known bytes per launch: {rows * rf * 4} ({rows * rf * 4 / 1024:.0f} KB)")
