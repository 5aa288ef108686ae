"""rocprofv3 PMC passes -> profiles/<name>.json with the HBM bytes per launch of the env-step kernel.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d A -- python3 tools/step_bench.py 4096 40
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d B -- python3 tools/step_bench.py 4096 40     (separate passes)
    python tools/pmc_traffic.py A B pf_step_kernel pointfoot_rough 4096 profiles/r2_step_kernel_hbm_pmc.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced reads, so it is doubled (an upper bound for
this kernel's mix of dword record loads and int16 gathers, whose width is uncalibrated); WRITE_SIZE is exact for streaming
stores.  bench.py reads the file for `roofline.traffic` and checks kernel / task / env count before using it."""
import collections, csv, glob, json, os, sys


def mean_counter(d, counter, pat):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    vals = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and pat in r["Kernel_Name"]:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    assert len(vals) == 1, f"{counter}: expected one kernel matching {pat!r}, got {list(vals)}"
    (name, v), = vals.items()
    v = v[len(v) // 4:]          # skip the warm-up launches
    return name, sum(v) / len(v), len(v)


if __name__ == "__main__":
    fetch_dir, write_dir, pat, task, n, out = sys.argv[1:7]
    req_dirs = sys.argv[7:]        # optional: directories of the request-size passes (TCC_EA0_RDREQ* / TCC_EA0_WRREQ*)
    name, fetch_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE", pat)
    _, write_kb, nw = mean_counter(write_dir, "WRITE_SIZE", pat)
    kernel = name.split("(")[0]
    rec = {"kernel": kernel, "task": task, "num_envs": int(n), "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
           "launches_averaged": [nf, nw],
           "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
           "note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 half-count correction on the read side (upper bound), separate --pmc passes"}
    if req_dirs:
        # The L2's memory-side requests by size (profiles/r4_fetch_size_gather_calibration.txt: on gfx950 FETCH_SIZE = 64 B x
        # TCC_EA0_RDREQ whatever the request size; a missing line is fetched as ONE 128-byte request, also under int16 gathers
        # that need one sector of it) -- the exact byte count the doubling rule approximates.
        c = {}
        for d in req_dirs:
            for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum",
                      "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
                try:
                    c[k] = mean_counter(d, k, pat)[1]
                except (AssertionError, ValueError, ZeroDivisionError):
                    pass
        rec["requests_per_launch"] = c
        rd = 128.0 * c["TCC_EA0_RDREQ_128B_sum"] + 64.0 * c["TCC_EA0_RDREQ_64B_sum"] + 32.0 * c["TCC_EA0_RDREQ_32B_sum"]
        wr = 64.0 * c["TCC_EA0_WRREQ_64B_sum"] + 32.0 * (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"])
        rec["read_bytes_per_launch"], rec["write_bytes_per_launch"] = rd, wr
        rec["traffic_doubling_rule_bytes_per_launch"] = rec["traffic_bytes_per_launch"]
        rec["traffic_bytes_per_launch"] = rd + wr
        rec["note"] = ("read = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B, write = 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B): the L2's "
                       "memory-side requests by size, separate --pmc passes; FETCH_SIZE / WRITE_SIZE of the same kernel beside them (2 x FETCH_SIZE "
                       "counts the 64-byte read requests twice: traffic_doubling_rule_bytes_per_launch)")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))
