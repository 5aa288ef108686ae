"""FETCH_SIZE calibration on the height scan's gather pattern (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated:
calibrate on a known byte count in your own access pattern").  Every launch lets R robots (32 lanes each) sample the 11 x 11
scan grid -- three int16 samples per point -- at base cells of a 512 MiB field no earlier launch has touched, so every sector a
launch needs comes from memory.  The host counts the distinct 64-byte sectors and 128-byte lines of each launch.
   rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/calib_gather.py
Cases (robots one 128-byte line apart): A = the twelve samples of a grid row inside ONE 64-byte sector, B = across the two sectors
of one line, C = across a line boundary, D = any alignment."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointfoot_amd import _abi
lib = _abi.load()
lib.pf_debug_calib_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
ROWS, COLS = 16640, 16384            # int16: 520 MiB, twice the Infinity Cache
R = 1 << 15                          # robots per launch
field = torch.randint(-2000, 2000, (ROWS, COLS), device="cuda:0", dtype=torch.int16)
sink = torch.zeros(4, device="cuda:0")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)


def footprint(base):
    rows = (base[:, 0:1] + np.arange(12)[None, :]).reshape(-1)                    # 12 field rows per robot (ix 0..10, +1)
    c0 = np.repeat(base[:, 1], 12)                                                # first sample column, last = c0 + 11
    lo, hi = (rows.astype(np.int64) * COLS + c0) * 2, (rows.astype(np.int64) * COLS + c0 + 11) * 2 + 1      # byte range of a row's samples
    out = []
    for g in (64, 128):
        a, b = lo // g, hi // g
        ids = np.concatenate([a, b[b != a]])
        out.append(len(np.unique(ids)) * g)
    return out


# launch k uses field rows [2064 k, 2064 (k + 1)): 129 bands of 16 rows, 255 robots per band, 64 columns (one 128-byte line) apart --
# no two robots share a line unless the case says so, and no launch touches a line an earlier one has read
def bases(case, k):
    band, slot = np.arange(R) // 255, np.arange(R) % 255      # (255 per band: case C's last row still ends inside the field row)
    bx = k * 2064 + band * 16
    off = {"A": 0,                                  # a grid row's 12 samples = bytes 0..23 of a line: one sector, one line
           "B": 26,                                 # bytes 52..75: two sectors of one line
           "C": 58,                                 # bytes 116..139: the last sector of a line and the first of the next (shared with the neighbour's row)
           "D": rng.integers(0, 53, R)}[case]       # any alignment, as under a walking robot
    return np.stack([bx, slot * 64 + off], 1).astype(np.int32)


k = 0
for case in ("A", "B", "C", "D"):
    for rep in range(2):
        b = bases(case, k); k += 1
        assert b[:, 0].max() + 12 < ROWS and b[:, 1].max() + 12 < COLS
        sec, lin = footprint(b)
        d = torch.from_numpy(b).to("cuda:0")
        assert lib.pf_debug_calib_gather(field.data_ptr(), COLS, d.data_ptr(), R, sink.data_ptr(), s) == 0
        torch.cuda.synchronize()
        print(f"launch {k - 1:2d} case {case}: samples {R * 121 * 3 * 2 / 1024:.0f} KB, distinct 64-B sectors {sec / 1024:.0f} KB, distinct 128-B lines {lin / 1024:.0f} KB")
