"""Register / spill / scratch / LDS figures of every kernel in the library, from the compiler's own metadata -- so that DESIGN.md
and profiles/ quote the shipped code object and cannot drift.  Compiles the two translation units for the
device only with exactly the flags of __graft_entry__.build_hip() and reads the .amdgpu_metadata notes of the assembly.
   python tools/kernel_resources.py [out.txt]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

def main():
    out = []
    flags = ge.build_flags()
    for src in ge.HIP_SOURCES:
        cmd = flags["common"] + flags["per_source"].get(src, []) + ["--cuda-device-only", "-S", "-o", "-", os.path.join(ge.CSRC, src)]
        asm = subprocess.run(cmd, cwd=ROOT, check=True, capture_output=True, text=True).stdout
        meta = asm[asm.index("amdhsa.kernels:"):]
        for blk in re.split(r"\n  - ", meta)[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            name = re.sub(r"\(.*", "", name)
            out.append((src, name, g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    lines = ["# kernel resources of pointfoot_amd/libpointfoot_hip.so (tools/kernel_resources.py: compiler metadata of the gfx950 code objects)",
             f"# {'kernel':88s} vgpr agpr vspill sgpr sspill scratchB staticLDS"]
    for src, name, v, a, vs, s, ss, sc, lds in sorted(out, key=lambda r: (r[0], r[1])):
        lines.append(f"{src[:-4]:8s} {name[:80]:80s} {v:>4s} {a:>4s} {vs:>6s} {s:>4s} {ss:>6s} {sc:>8s} {lds:>9s}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)

if __name__ == "__main__":
    main()
