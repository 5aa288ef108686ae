"""Timeline of ONE policy step of the rollout, from a rocprofv3 kernel trace of bench.py:
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 10 --warmup 3 --no_cpu_baseline
   python tools/rollout_timeline.py DIR
For every pair of consecutive env-step launches less than 400 us apart (i.e. inside a rollout) the kernels between them are lined up by
position; prints mean duration of each and the mean idle gap in front of it.  The step period minus the kernels is what dependent
launches cost on one stream."""
import collections, csv, glob, os, sys

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n

def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    steps = [i for i, r in enumerate(rows) if "pf_step_kernel" in r[2]]
    seqs = collections.defaultdict(list)
    for a, b in zip(steps, steps[1:]):
        if rows[b][0] - rows[a][0] > 400_000:
            continue
        win = rows[a:b]
        key = tuple(short(r[2]) for r in win)
        seqs[key].append((win, rows[b][0]))
    key, best = max(seqs.items(), key=lambda kv: len(kv[1]))
    print(f"{len(best)} policy steps with the common kernel sequence ({len(seqs)} distinct sequences seen)")
    n = len(best)
    period = sum(nxt - win[0][0] for win, nxt in best) / n / 1e3
    tot_k = tot_g = 0.0
    for i, name in enumerate(key):
        dur = sum(win[i][1] - win[i][0] for win, _ in best) / n / 1e3
        nxt_start = [(win[i + 1][0] if i + 1 < len(key) else nxt) for win, nxt in best]
        gap = sum(s - win[i][1] for s, (win, _) in zip(nxt_start, best)) / n / 1e3
        tot_k += dur; tot_g += gap
        print(f"  {dur:7.1f} us  then idle {gap:5.1f} us   {name}")
    print(f"period {period:.1f} us = kernels {tot_k:.1f} + idle {tot_g:.1f}")

if __name__ == "__main__":
    main()
