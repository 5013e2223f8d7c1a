"""Static instruction mix between consecutive phase marks (s_memtime) of a kernel in a -DMBRL_PHASE_TIMING assembly.
usage: python tools/isa_phases.py file.s <substring of the kernel symbol>"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def cls(op):
    if op.startswith("v_accvgpr"): return "agpr"
    if op.endswith("_f64") or "_f64_" in op:
        return "cmp64" if op.startswith("v_cmp") else "fp64"
    if op.startswith("v_mov"): return "vmov"
    if op.startswith("v_cndmask"): return "cnd"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_mov"): return "smov"
    if op.startswith("s_"): return "salu"
    return "oth"
seg, segs, labels = collections.Counter(), [], 0
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith((".", ";", "/")): continue
    if t.endswith(":"): seg["labels"] += 1; continue
    op = t.split()[0]
    if op == "s_memtime":
        segs.append(seg); seg = collections.Counter(); continue
    seg[cls(op)] += 1
segs.append(seg)
keys = ["fp64", "cmp64", "agpr", "vmov", "cnd", "valu", "smov", "salu", "lds", "wait", "br", "oth", "labels"]
print("seg  total " + " ".join("%6s" % k for k in keys))
for i, s in enumerate(segs):
    tot = sum(v for k, v in s.items() if k != "labels")
    print("%3d %6d " % (i, tot) + " ".join("%6d" % s[k] for k in keys))
