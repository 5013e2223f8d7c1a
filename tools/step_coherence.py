"""Would ordering the episodes by control flow make a batched rollout's step launches faster?  Rolls B episodes from
reset-like states under random controls and, at several time steps, times the step launch on the episodes in their natural
order, ordered by the PREVIOUS step's launch-plan key (what a trainer could know in time) and by the CURRENT step's key
(the bound).  The order is applied by permuting the inputs, so the timings are of the step kernel alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine

env = sys.argv[1] if len(sys.argv) > 1 else "HalfCheetahEnv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
e = engine.FdEngine(env, 0, "cuda:0")
g = torch.Generator(device="cuda"); g.manual_seed(0)
q = (torch.rand((B, e.nq), generator=g, device="cuda", dtype=torch.float64) - 0.5) * 0.2
v = torch.randn((B, e.nv), generator=g, device="cuda", dtype=torch.float64) * 0.1
if env in ("HopperEnv", "Walker2dEnv"):
    q[:, 1] += 1.25
up = lambda x: (x + 255) & ~255

def keys(q, v, a):
    e.fd_plan(q, v, a)
    buf = e._stream_plan().buf
    return buf[256:256 + 8 * B].view(torch.int64).clone()

def timed(q, v, a, reps=7):
    t = []
    for _ in range(reps):
        s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); e.step(q, v, a); f.record(); torch.cuda.synchronize()
        t.append(s.elapsed_time(f))
    return sorted(t)[len(t) // 2]

prev = None
for t in range(40):
    a = torch.clamp(torch.randn((B, e.nu), generator=g, device="cuda", dtype=torch.float64) * 0.5, -1, 1)
    if t in (1, 5, 10, 20, 30, 39):
        kc = keys(q, v, a)
        pc = torch.argsort(kc)
        pp = torch.argsort(prev)
        rnd = torch.randperm(B, device="cuda")
        print("t %2d: natural %.0f us   random order %.0f us   by previous step's key %.0f us   by this step's key %.0f us" % (
            t, 1e3 * timed(q, v, a), 1e3 * timed(q[rnd].contiguous(), v[rnd].contiguous(), a[rnd].contiguous()),
            1e3 * timed(q[pp].contiguous(), v[pp].contiguous(), a[pp].contiguous()),
            1e3 * timed(q[pc].contiguous(), v[pc].contiguous(), a[pc].contiguous())), flush=True)
    prev = keys(q, v, a)
    q, v, r, st = e.step(q, v, a)
