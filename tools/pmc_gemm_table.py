"""profiles/rN_gemm_pmc.txt's table from the raw per-pass listing of tools/pmc_bench.sh (kernel name line, then `COUNTER value (calls n)`
lines, one block per kernel and pass):   python tools/pmc_gemm_table.py gpurun_out/r3_gemm_pmc_raw.txt"""
import re, sys

def main():
    cur, rows, order = None, {}, []
    for line in open(sys.argv[1]):
        m = re.match(r"^\s+(\w+)\s+(\d+)\s+\(calls (\d+)\)", line)
        if m and cur:
            rows[cur][m.group(1)] = float(m.group(2)); rows[cur]["calls"] = int(m.group(3))
        elif line.startswith("void ") or line.startswith("pf_"):
            cur = line.strip()
            if cur not in rows:
                rows[cur] = {}; order.append(cur)
    print(f"{'kernel':62s} {'calls':>5s} {'kcycles':>8s} {'mfma_busy':>10s} {'valu/mfma':>10s} {'lds_conflict':>13s} {'wait':>6s} {'stall':>6s} {'MFMA/launch':>14s} {'VALU/launch':>13s}")
    for k in order:
        r = rows[k]
        if "GRBM_GUI_ACTIVE" not in r or "SQ_INSTS_MFMA" not in r:
            continue
        kc = r["GRBM_GUI_ACTIVE"] / 8.0
        print(f"{k[:62]:62s} {r['calls']:5d} {kc / 1e3:7.1f}k {r['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * kc):10.3f} {r['SQ_INSTS_VALU'] / r['SQ_INSTS_MFMA']:10.2f} "
              f"{r['SQ_LDS_BANK_CONFLICT'] / r['SQ_LDS_IDX_ACTIVE']:13.3f} {r['SQ_WAIT_ANY'] / r['SQ_WAVE_CYCLES']:6.2f} {r['SQ_WAIT_INST_ANY'] / r['SQ_WAVE_CYCLES']:6.2f} "
              f"{r['SQ_INSTS_MFMA']:14.0f} {r['SQ_INSTS_VALU']:13.0f}")

if __name__ == "__main__":
    main()
