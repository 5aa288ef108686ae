"""Where does a kernel spill?  Compiles pf_env.hip for the device with line tables and counts the scratch stores / loads of one
kernel by the source line the compiler attributes them to.
   python tools/spill_sites.py <mangled-name-prefix> [extra hipcc flags ...]     e.g.  _Z14pf_step_kernelILi17ELi12ELi4ELi2ELi1ELi32ELi1ELi8ELi0"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

def main():
    prefix, extra = sys.argv[1], sys.argv[2:]
    fl = ge.build_flags()
    src = "pf_env_quad.hip" if "ILi17E" in prefix else "pf_env.hip"
    per = [] if os.environ.get("SCHED") == "none" else fl["per_source"].get(src, [])
    cmd = fl["common"] + per + extra + ["--cuda-device-only", "-gline-tables-only", "-S", "-o", "-", os.path.join(ge.CSRC, src)]
    asm = subprocess.run(cmd, cwd=ROOT, check=True, capture_output=True, text=True).stdout.split("\n")
    files = {}
    for l in asm:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m: files[int(m.group(1))] = os.path.basename(m.group(2))
    starts = [i for i, l in enumerate(asm) if l.startswith(prefix) and l.rstrip().split(":")[0].startswith(prefix) and ":" in l]
    if not starts:
        sys.exit("no kernel with that prefix: " + ", ".join(sorted({l.split(":")[0] for l in asm if l.startswith("_Z14pf_step_kernel")})))
    start = starts[0]
    end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
    cur, st, ld = None, collections.Counter(), collections.Counter()
    for l in asm[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        if "scratch_store" in l: st[cur] += 1
        if "scratch_load" in l: ld[cur] += 1
    print(asm[start].split(":")[0])
    print(f"instructions {end - start}, scratch stores {sum(st.values())}, scratch loads {sum(ld.values())}")
    for name, c in (("stores", st), ("loads", ld)):
        print(name + ":", ", ".join(f"{f}:{n} x{k}" for (f, n), k in sorted(c.items(), key=lambda x: -x[1])[:30]))

if __name__ == "__main__":
    main()
