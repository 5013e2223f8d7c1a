"""What ordering the units BEFORE the plan launch could be worth (not a test): mbrl_fd_plan on the S2 batch as it comes,
against the same launch on the batch permuted into the order the plan itself ends up with (the best any cheap pre-key
could do: waves of the plan launch then hold 64 like units instead of 64 unrelated ones).
usage: python tools/plan_presort_bound.py [env] [N]"""
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
q, v, c = workloads.s2(e, N, seed=0)
up = lambda x: (x + 255) & ~255


def perm_of(plan):
    off = 256 + 2 * up(8 * N) + up(4 * N)
    return plan.buf[off:off + 4 * N].view(torch.int32).long().clone()


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


plan, out = e.new_plan(N), e.new_outputs(N)
t0 = timed(lambda: e.fd_plan(q, v, c, out=out, plan=plan))
t0p = timed(lambda: e.fd_planned(q, v, c, out, plan=plan))
p = perm_of(plan)
assert sorted(p.tolist()) == list(range(N))
qs, vs, cs = q[p].contiguous(), v[p].contiguous(), c[p].contiguous()
t1 = timed(lambda: e.fd_plan(qs, vs, cs, out=out, plan=plan))
t1p = timed(lambda: e.fd_planned(qs, vs, cs, out, plan=plan))
print("%s N=%d: plan launch %.3f ms as the batch comes, %.3f ms on the batch in plan order (planned launch %.3f / %.3f ms)"
      % (env, N, t0, t1, t0p, t1p))
