"""Scratch loads / stores of one kernel variant by source line (development aid).
usage: python tools/spill_lines.py <TG_VARIANTS mask> <substring of the mangled kernel name>"""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/spill_lines.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DTG_VARIANTS=" + sys.argv[1],
                "-gline-tables-only", "-S", "--cuda-device-only", "-o", out, "admm_kernels.hip"] + sys.argv[3:],
               cwd=os.path.join(root, "trajectory_generation_amd", "csrc"), stderr=subprocess.DEVNULL, check=True)
src = open(out).read().split("\n")
start = [i for i, l in enumerate(src) if re.match(r"^_ZN2tg11admm_kernel\S*" + re.escape(sys.argv[2]) + r"\S*:", l)][0]
end = [i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end")][0]
cur = None
ld, st = collections.Counter(), collections.Counter()
for l in src[start:end]:
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)\s+(\d+)", l)
    if m: cur = int(m.group(1))
    if "scratch_load" in l: ld[cur] += 2 if "dwordx2" in l else 1
    if "scratch_store" in l: st[cur] += 2 if "dwordx2" in l else 1
print("dwords loaded by source line ", sorted(ld.items()), sum(ld.values()))
print("dwords stored by source line", sorted(st.items()), sum(st.values()))
for l in src[start:end + 200]:
    if re.search(r"\.(vgpr_spill_count|private_segment_fixed_size):", l): print(l.strip())
