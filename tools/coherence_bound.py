"""What perfect wave coherence would be worth (not a test): the planned FD launch on the S2 batch as it is, against the
same launch on a batch in which every wave pair holds COPIES of one unit (a 1-in-K sample of the batch, each unit repeated
K times, K = units per wave pair) -- same average work per unit, no union of different units' control flow.
usage: python tools/coherence_bound.py [env] [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import workloads
from mbrl_amd.engine import FdEngine

env = sys.argv[1] if len(sys.argv) > 1 else "HalfCheetahEnv"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 163840
e = FdEngine(env)
K = {"HalfCheetahEnv": 5, "Walker2dEnv": 5, "HopperEnv": 8, "SwimmerEnv": 10}.get(env, 5)
q, v, c = workloads.s2(e, N, seed=0)


def timed(q, v, c, reps=5):
    out = e.fd_jacobians(q, v, c, planned=True)
    plan = e.last_plan if hasattr(e, "last_plan") else None
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        a.record()
        e.fd_jacobians(q, v, c, out=out, planned=True)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


t_batch = timed(q, v, c)
for phase in range(K):
    idx = torch.arange(phase, N, K, device=q.device)[: N // K].repeat_interleave(K)
    t_rep = timed(q[idx].contiguous(), v[idx].contiguous(), c[idx].contiguous())
    print("%s N=%d: batch %.3f ms (plan + planned), %d-fold copies of sample %d: %.3f ms -> coherence is worth %.1f %%" % (
        env, N, t_batch, K, phase, t_rep, 100.0 * (t_batch / t_rep - 1.0)), flush=True)
    if phase >= 1:
        break
