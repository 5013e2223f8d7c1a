"""Instruction mix of the code that FOLLOWS each `; MBRL_MARK name` comment (a -DMBRL_ASM_MARKS assembly) up to the next
mark, in layout order -- the fall-through (hot) path; rare blocks live out of line after `s_endpgm`-side labels.
usage: python tools/isa_marks.py file.s <substring of the kernel symbol> [dump=<mark name>]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
dump = next((a[5:] for a in sys.argv[3:] if a.startswith("dump=")), None)
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
segs, name, seg = [], "(prologue)", collections.Counter()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; MBRL_MARK (\S+)", t)
    if m:
        segs.append((name, seg)); name, seg = m.group(1), collections.Counter(); continue
    if not t or t.startswith((".", ";", "/")): continue
    if t.endswith(":"): seg["labels"] += 1; continue
    if dump == name: print("   ", t.split(";")[0].rstrip())
    seg[cls(t.split()[0])] += 1
segs.append((name, seg))
keys = ["fp64", "cmp64", "agpr", "vmov", "cnd", "valu", "smov", "salu", "lds", "wait", "br", "oth", "labels"]
print("%-20s total " % "after mark" + " ".join("%6s" % k for k in keys))
for name, s in segs:
    tot = sum(v for k, v in s.items() if k != "labels")
    print("%-20s %5d " % (name, tot) + " ".join("%6d" % s[k] for k in keys))
