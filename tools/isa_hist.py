"""Static instruction histogram of one kernel in a device assembly file (hipcc -S --cuda-device-only).
usage: python tools/isa_hist.py file.s <substring of the kernel symbol> [top]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end + 1]
ops = collections.Counter()
cls = collections.Counter()
for l in body:
    t = l.strip()
    if not t or t.startswith((".", ";", "/")) or t.endswith(":"):
        continue
    op = t.split()[0]
    ops[op] += 1
    if op.startswith("v_accvgpr"): k = "agpr copy"
    elif re.match(r"v_(fma|fmac|mul|add|min|max|rcp|rsq|sqrt|div|ldexp|frexp|trig|fract|rndne|cmp\w*_f64)\w*_f64", op) or op.endswith("_f64"): k = "fp64 valu"
    elif op.startswith("v_mov") or op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): k = "v_mov/lane"
    elif op.startswith("v_cndmask"): k = "v_cndmask"
    elif op.startswith("v_"): k = "other valu"
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): k = "vmem"
    elif op.startswith("s_waitcnt"): k = "s_waitcnt"
    elif op.startswith(("s_cbranch", "s_branch")): k = "branch"
    elif op.startswith("s_load"): k = "s_load"
    elif op.startswith("s_nop"): k = "s_nop"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    cls[k] += 1
tot = sum(cls.values())
print("kernel", lines[start][:90], "instructions", tot)
for k, n in cls.most_common():
    print("  %-12s %6d  %5.1f %%" % (k, n, 100.0 * n / tot))
for op, n in ops.most_common(top):
    print("    %-28s %6d" % (op, n))
