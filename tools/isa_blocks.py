"""Basic blocks of a kernel in layout order: label, instruction mix, marks seen, where it branches.
usage: python tools/isa_blocks.py file.s <kernel symbol substring> [min_instrs]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def cls(op):
    if op.startswith("v_accvgpr"): return "agpr"
    if op.endswith("_f64") or "_f64_" in op: return "cmp64" if op.startswith("v_cmp") else "fp64"
    if op.startswith("v_mov"): return "vmov"
    if op.startswith("v_cndmask"): return "cnd"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_mov"): return "smov"
    if op.startswith("s_"): return "salu"
    return "oth"
blocks, cur = [], dict(label="entry", c=collections.Counter(), marks=[], br=[])
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r"; MBRL_MARK (\S+)", t)
    if m: cur["marks"].append(m.group(1)); continue
    if not t or t.startswith((".loc", ".cfi", ";", "/", ".p2align")): continue
    m = re.match(r"^(\.LBB\S+):", t)
    if m:
        blocks.append(cur); cur = dict(label=m.group(1), c=collections.Counter(), marks=[], br=[]); continue
    if t.startswith("."): continue
    op = t.split()[0]
    cur["c"][cls(op)] += 1
    if op.startswith(("s_cbranch", "s_branch")): cur["br"].append(op.replace("s_cbranch_", "").replace("s_branch", "jmp") + ">" + t.split()[-1].replace(".LBB", ""))
blocks.append(cur)
keys = ["fp64", "cmp64", "agpr", "vmov", "cnd", "valu", "smov", "salu", "lds", "wait"]
print("%-12s %5s " % ("block", "n") + " ".join("%5s" % k for k in keys) + "  marks / branches")
for b in blocks:
    n = sum(b["c"].values())
    if n >= mn or b["marks"]:
        print("%-12s %5d " % (b["label"].replace(".LBB", ""), n) + " ".join("%5d" % b["c"][k] for k in keys) + "  " + ",".join(b["marks"]) + " " + " ".join(b["br"]))
