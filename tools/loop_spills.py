"""Per kernel variant: the loops of the ISA (spans closed by a backward branch) with their counts
of MFMA, barrier and scratch instructions -- shows whether register spills sit in the hot loop.
usage: python tools/loop_spills.py file.s [substring of the kernel name]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
fn, start = None, 0
funcs = {}
for i, l in enumerate(src):
    m = re.match(r"^(_ZN2tg11admm_kernel\S*):", l)
    if m: fn = m.group(1); funcs[fn] = [i, None]
    if fn and l.startswith(".Lfunc_end") and funcs[fn][1] is None: funcs[fn][1] = i
for fn, (a, b) in funcs.items():
    if want not in fn or b is None: continue
    labels = {}
    for i in range(a, b + 1):
        m = re.match(r"^(\.LBB\d+_\d+):", src[i])
        if m: labels[m.group(1)] = i
    print(fn[19:60])
    for i in range(a, b + 1):
        m = re.match(r"^\s*s_c?branch\S*\s+(\.LBB\d+_\d+)", src[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lo = labels[m.group(1)]
            body = src[lo:i + 1]
            c = lambda pat: sum(1 for x in body if re.search(pat, x))
            if c("v_mfma") or c("s_barrier"):
                print(f"  loop lines {lo - a}-{i - a} ({i - lo} lines): mfma {c('v_mfma')} barrier {c('s_barrier')} "
                      f"scratch_load {c('scratch_load')} scratch_store {c('scratch_store')} valu_f64 {c('v_(fma|mul|add|max|min)_f64')}")
