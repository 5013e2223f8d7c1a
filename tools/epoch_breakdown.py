"""Where one training epoch's GPU time goes (open loop, B episodes x H steps): H step launches, the linearisation of
the B*H stored (s, a) (plan + perturbed rollouts), the backward scan.  usage: python tools/epoch_breakdown.py [B] [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine, rollout, workloads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 40
e = engine.FdEngine("HalfCheetahEnv", 0, "cuda:0")
q, v, _ = workloads.s1(e, B, seed=0)
s0 = torch.cat([q, v], dim=1)
g = torch.Generator(device="cpu").manual_seed(1)
acts = torch.clamp(0.5 * torch.randn((B, H, e.nu), generator=g, dtype=torch.float64), -1, 1).cuda()
c = torch.ones((B, H), dtype=torch.float64, device="cuda")

def ev():
    return torch.cuda.Event(enable_timing=True)

for rep in range(3):
    t = [ev() for _ in range(5)]
    t[0].record()
    states, rewards, status = rollout.rollout(e, s0, acts)
    t[1].record()
    s = states[:, :H].reshape(B * H, e.n)
    qq, vv, aa = s[:, :e.nq].contiguous(), s[:, e.nq:].contiguous(), acts.reshape(B * H, e.nu)
    out = e.fd_plan(qq, vv, aa)
    t[2].record()
    e.fd_planned(qq, vv, aa, out)
    t[3].record()
    da, ds0 = e.vjp_scan(out, c, "average")
    t[4].record()
    torch.cuda.synchronize()
    names = ["%d step launches" % H, "plan (nominal rollouts + sort)", "perturbed rollouts", "backward scan"]
    print("rep %d: " % rep + "  ".join("%s %.3f ms" % (n, t[i].elapsed_time(t[i + 1])) for i, n in enumerate(names)),
          " bad %d" % int((out["status"] != 0).sum()), flush=True)
