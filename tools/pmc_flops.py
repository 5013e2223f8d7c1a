"""Executed fp64 lane-flops of the dominant kernel from the instruction-mix PMC pass of tools/pmc_pass.sh:
64 x (2 FMA + ADD + MUL + TRANS) fp64 VALU instructions per launch -- an UPPER bound on useful work (an instruction
issued with a partial EXEC mask counts all 64 lanes).  bench.py puts it next to the algorithmic count.
usage: python tools/pmc_flops.py gpurun_out/pmc_<tag> <units> <fd_mode> <out.json>"""
import json, re, sys
root, units, fd_mode, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
v = {}
for line in open(root + "/summary.txt"):
    m = re.match(r"(SQ_\w+)\s+n=\S+\s+mean=(\S+)", line)
    if m:
        v[m.group(1)] = float(m.group(2))
fma, add, mul, tr = (v["SQ_INSTS_VALU_%s_F64" % k] for k in ("FMA", "ADD", "MUL", "TRANS"))
rec = {"units": units, "fd_mode": fd_mode, "kernel": "rollout_kernel<MdlHalfCheetah,false,false> (mbrl_fd_jacobians_planned)",
       "waves": v.get("SQ_WAVES"), "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"),
       "fma_f64": fma, "add_f64": add, "mul_f64": mul, "trans_f64": tr,
       "lane_flops_per_launch": 64.0 * (2 * fma + add + mul + tr),
       "note": "wave-level instruction counts x 64 lanes; partial-EXEC instructions count in full (upper bound)"}
json.dump(rec, open(out, "w"), indent=1)
print(rec)
