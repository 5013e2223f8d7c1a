"""What should the launch plan sort the units by?  Computes the per-mj_step control-flow trace of the nominal rollouts
on the host (tests/host_twin.cpp, twin_trace), builds candidate unit orders from it, writes each into the plan buffer
in place of the kernel's own order and times the planned launch.  An experiment, not a test.
usage: python tools/key_probe.py [env] [N]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine, workloads, modelspec

env = sys.argv[1] if len(sys.argv) > 1 else "HopperEnv"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 163840
e = engine.FdEngine(env, 0, "cuda:0")
q, v, c = workloads.s2(e, N, seed=0, horizon=40)
out = e.fd_plan(q, v, c)
plan = e._stream_plan()
up = lambda x: (x + 255) & ~255
perm_off = 256 + 2 * up(8 * N) + up(4 * N)
perm_view = plan.buf[perm_off:perm_off + 4 * N].view(torch.int32)
own = perm_view.clone()

def timed(reps=5):
    t = []
    for _ in range(reps):
        a, b = (torch.cuda.Event(enable_timing=True) for _ in range(2))
        a.record(); e.fd_planned(q, v, c, out); b.record()
        torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    return sorted(t)[len(t) // 2]

e.fd_planned(q, v, c, out); torch.cuda.synchronize()
ref = {k: out[k].clone() for k in ("dsds", "dsda", "drds", "drda")}
print(env, N, "units")
print("%-52s %.3f ms" % ("kernel's own key", timed()), flush=True)

tw = C.CDLL(os.path.join(ROOT, "tests", "_build", "libhost_twin.so"))
steps = int(modelspec.MODELS[env]["frame_skip"])
env_id = list(modelspec.MODELS).index(env)
tr = np.zeros((N, steps), dtype=np.uint32)
P = lambda a: a.ctypes.data_as(C.c_void_p)
qh, vh, ch = (np.ascontiguousarray(x.cpu().numpy()) for x in (q, v, c))
tw.twin_trace(C.c_int(env_id), C.c_int(steps), C.c_int64(N), P(qh), P(vh), P(ch), P(tr), C.c_int(16))
ps = (tr & 15).astype(np.int64)
ls = ((tr >> 4) & 15).astype(np.int64)
rows = (tr >> 8).astype(np.int64)          # candidates | limits | pairs
print("steps", steps, "mean passes/step %.2f" % ps.mean(), "distinct row masks", len(np.unique(rows)), flush=True)

def lex(cols):
    """ascending lexicographic order, cols[0] most significant"""
    return np.lexsort(tuple(reversed(cols)))

def use(order, name):
    # the kernel walks slots from N-1 down: the order's direction does not matter for coherence, keep ascending
    perm_view.copy_(torch.from_numpy(order.astype(np.int32)).to(perm_view.device))
    ms = timed()
    same = all(torch.equal(out[k], ref[k]) for k in ref)
    print("%-52s %.3f ms%s" % (name, ms, "" if same else "  OUTPUTS DIFFER"), flush=True)

T = steps
pc = lambda x: np.array([bin(int(y)).count("1") for y in np.unique(x)])[np.searchsorted(np.unique(x), x)]
union = np.bitwise_or.reduce(rows, axis=1)
h1 = np.bitwise_or.reduce(rows[:, :T // 2], axis=1)
h2 = np.bitwise_or.reduce(rows[:, T // 2:], axis=1)
tot = ps.sum(1)
ncon = (ps > 0).sum(1)
P_ = [ps[:, k] for k in range(T)]
PR = [x for k in range(T) for x in (ps[:, k], rows[:, k])]
use(np.arange(N), "unsorted")
use(lex(P_), "passes, time-major")
use(lex([tot] + P_), "total passes, then passes")
use(lex([union] + P_), "union of rows, then passes")
use(lex([ncon, union] + P_), "constrained steps, union, passes")
use(lex([tot, union] + P_), "total passes, union, passes")
use(lex([union, tot] + P_), "union, total passes, passes")
use(lex([union, ncon, tot] + P_), "union, constrained steps, total passes, passes")
use(lex([tot, union] + PR), "total passes, union, per step (passes, rows)")
use(lex([pc(union), union] + P_), "popcount(union), union, passes")
use(lex([h1, h2] + P_), "union of first half, of second half, passes")
use(lex([union, h1, h2] + P_), "union, halves, passes")
use(lex([tot // 4, union] + P_), "total passes / 4, union, passes")
use(lex([tot // 8, union, tot] + P_), "total passes / 8, union, total, passes")
use(lex([ncon // 4, union, tot] + P_), "constrained steps / 4, union, total, passes")
def snake(cols, base):
    """reflected (boustrophedon) order: a digit runs backwards whenever the digits above it sum to an odd number, so
    neighbours across a carry differ in one digit only"""
    out, par = [], np.zeros(N, dtype=np.int64)
    for c in cols:
        c = np.minimum(c, base - 1)
        out.append(np.where(par & 1, base - 1 - c, c))
        par = par + c
    return out
L_ = [ls[:, k] for k in range(T)]
use(lex(P_ + L_), "passes, then line evals (the kernel's short-rollout key)")
use(lex([x for k in range(T) for x in (ps[:, k], ls[:, k])]), "per step (passes, line evals)")
use(lex(P_ + [rows[:, k] for k in range(T)]), "passes, then rows per step")
use(lex(P_ + L_ + [rows[:, k] for k in range(T)]), "passes, line evals, rows per step")
use(lex(list(reversed(P_)) + list(reversed(L_))), "time-reversed passes, line evals")
use(lex(snake(P_, 8) + L_), "snake(passes), line evals")
use(lex(snake(P_ + L_, 8)), "snake(passes, line evals)")
use(lex([ncon] + P_ + L_), "constrained steps, passes, line evals")
use(lex([ps.max(1)] + P_ + L_), "max passes, passes, line evals")
cost = ps + 0.3 * ls
use(lex([np.round(cost.sum(1)).astype(np.int64)] + P_ + L_), "estimated cost, passes, line evals")
use(lex([np.round(cost.sum(1) / 4).astype(np.int64)] + P_ + L_), "estimated cost / 4, passes, line evals")
perm_view.copy_(own)
print("%-52s %.3f ms" % ("kernel's own key again", timed()))
