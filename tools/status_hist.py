"""Status-bit histogram of an FD launch on S2 states, per env (MBRL_ST_*: 1 nonfinite, 2 noconverge, 4 overflow, 8 degenerate)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine, workloads
for env in ("HopperEnv", "Walker2dEnv", "HalfCheetahEnv"):
    e = engine.FdEngine(env)
    N = 16384
    q, v, c = workloads.s2(e, N, seed=0)
    o = e.fd_jacobians(q, v, c, planned=False)
    st = o["status"].cpu()
    vals, cnt = torch.unique(st, return_counts=True)
    print(env, {int(a): int(b) for a, b in zip(vals, cnt)}, "torso z of flagged units: min %.2f max %.2f" % (
        (float(q[st.cuda() != 0][:, 1].min()), float(q[st.cuda() != 0][:, 1].max())) if int((st != 0).sum()) else (0, 0)), flush=True)
