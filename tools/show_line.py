import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print("value %.4g ms_per_step %.4f kernel_ms %.4f steps %d" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["steps"]))
