"""Summarise rocprofv3 counter CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")[:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(root, "summary.txt"), "w") as fh:
    for k, d in acc.items():
        if "rollout_kernel" not in k: continue
        for c, v in sorted(d.items()):
            line = "%-62s %-28s n=%-4d mean=%.6g" % (k, c, len(v), sum(v) / len(v))
            print(line); fh.write(line + "\n")
