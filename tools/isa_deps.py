"""Static dependency-distance histogram of the VALU instructions in a range of lines of a device assembly listing:
for each VALU instruction, how many instructions back was the latest VALU producer of one of its source registers.
usage: python tools/isa_deps.py file.s first_line last_line"""
import re, sys, collections
path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lines = open(path).read().split("\n")[a - 1:b]
def regs(tok):
    out = []
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            out += [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
        else:
            out.append((m.group(4), int(m.group(5))))
    return out
last = {}
n = 0
hist = collections.Counter()
kinds = collections.Counter()
for l in lines:
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    op, _, rest = t.partition(" ")
    n += 1
    if not op.startswith("v_"):
        if op.startswith("ds_read") or op.startswith("global_load"):
            ops = [x.strip() for x in rest.split(",")]
            for r in regs(ops[0]): last[r] = (n, "mem")
        continue
    ops = [x.strip() for x in rest.split(",")]
    nd = 1
    if op.startswith("v_cmp") and not op.endswith("_e64"): nd = 0
    if op.startswith("v_div_scale"): nd = 2
    dst = sum((regs(o) for o in ops[:nd]), [])
    src = sum((regs(o) for o in ops[nd:]), [])
    if op.startswith(("v_fmac", "v_mac")): src += dst
    d = min((n - last[r][0] for r in src if r in last and last[r][1] == "valu"), default=99)
    hist[min(d, 6)] += 1
    if d == 1: kinds[op] += 1
    for r in dst: last[r] = (n, "valu")
tot = sum(hist.values())
print("VALU instructions", tot, "of", n)
for d in sorted(hist): print("  distance %s%d: %5d  %4.1f %%" % (">=" if d == 6 else "", d, hist[d], 100.0 * hist[d] / tot))
print("distance-1 consumers:", kinds.most_common(8))
