"""How much of the planned kernel's time is still control-flow divergence?  Times it on the bench batch and on a batch
made of the same units each repeated 64 times (every wave then runs copies of two or three units only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine, workloads

def time_planned(e, q, v, c, reps=5):
    out = e.fd_plan(q, v, c)
    e.fd_planned(q, v, c, out)
    t = []
    for _ in range(reps):
        a, b, d = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(); e.fd_plan(q, v, c, out=out); b.record(); e.fd_planned(q, v, c, out); d.record()
        torch.cuda.synchronize()
        t.append((a.elapsed_time(b), b.elapsed_time(d)))
    t.sort(key=lambda x: x[1])
    return t[len(t) // 2]

e = engine.FdEngine(sys.argv[1] if len(sys.argv) > 1 else "HalfCheetahEnv", 0, "cuda:0")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 163840
q, v, c = workloads.s2(e, N, seed=0, horizon=40)
def time_single(q, v, c, reps=5):
    out = e.fd_jacobians(q, v, c, planned=False)
    t = []
    for _ in range(reps):
        a, b = (torch.cuda.Event(enable_timing=True) for _ in range(2))
        a.record(); e.fd_jacobians(q, v, c, out=out, planned=False); b.record()
        torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    return sorted(t)[len(t) // 2]

print(e.env, N, "units; single launch: bench batch %.3f ms" % time_single(q, v, c))
print("bench batch        plan %.3f ms  kernel %.3f ms" % time_planned(e, q, v, c))
rep = lambda x: x[: N // 64].repeat_interleave(64, dim=0).contiguous()
print("64x repeated units plan %.3f ms  kernel %.3f ms" % time_planned(e, rep(q), rep(v), rep(c)))
rep2 = lambda x: x[: N // 64].repeat(64, 1).contiguous()
print("single launch, 64x repeated units %.3f ms" % time_single(rep(q), rep(v), rep(c)))
print("same units, tiled  plan %.3f ms  kernel %.3f ms" % time_planned(e, rep2(q), rep2(v), rep2(c)))
