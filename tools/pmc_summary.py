"""Summarise rocprofv3 counter CSVs for the FD launches (the dispatches with the largest grid)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "rollout_kernel" in r.get("Kernel_Name", "")]
    if not rows: continue
    gmax = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == gmax:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(root, "summary.txt"), "w") as fh:
    for c, v in sorted(acc.items()):
        line = "%-28s n=%-4d mean=%.6g" % (c, len(v), sum(v) / len(v))
        print(line); fh.write(line + "\n")
    w = acc.get("SQ_WAVES", [0])[0]
    if w:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"):
            if c in acc:
                line = "per-wave %-24s %.1f" % (c, sum(acc[c]) / len(acc[c]) / w)
                print(line); fh.write(line + "\n")
