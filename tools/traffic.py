"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_pass.sh into profiles/<tag>_traffic.json.
usage: python tools/traffic.py gpurun_out/pmc_<tag> <units> <out.json>"""
import json, re, sys
root, units, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
v = {}
for line in open(root + "/summary.txt"):
    m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\S+\s+mean=(\S+)", line)
    if m:
        v[m.group(1)] = float(m.group(2))
# rocprofv3 reports both in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM): x2
rec = {"units": units, "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
       "hbm_bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0,
       "note": "separate --pmc passes; FETCH_SIZE doubled (gfx950 correction); inputs are read 8 B/lane, for which "
               "the guide calls the correction uncalibrated -- the read side is <8% of the bytes either way"}
json.dump(rec, open(out, "w"), indent=1)
print(rec)
