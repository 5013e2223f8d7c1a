"""Ad-hoc timing probe (not collected by pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import mbrl_amd
from mbrl_amd.engine import FdEngine
import util
for env in (sys.argv[1:] or ["HalfCheetahEnv"]):
    e = FdEngine(env)
    N = 16384
    q, v, c = util.reset_like(env, N, 0, e.nq, e.nu)
    q, v, c = (torch.as_tensor(x, device="cuda") for x in (q, v, c))
    out = e.fd_jacobians(q, v, c); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): out = e.fd_jacobians(q, v, c, out=out)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print(env, "N", N, "ms", dt * 1e3, "J/s %.3e" % (N / dt), "bad status", int((out["status"] != 0).sum()), flush=True)
