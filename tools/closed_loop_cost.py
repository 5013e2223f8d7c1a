"""Closed-loop linearisation at the bench size: cost of evaluating the policy on the N*(nq+nv) perturbed states
(FdEngine.perturbed_policy_actions, a torch GEMM chain) next to the FD launches themselves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine, workloads
e = engine.FdEngine("HalfCheetahEnv")
N = 163840
q, v, c = workloads.s2(e, N, seed=0)
pol = torch.nn.Sequential(torch.nn.Linear(18, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                          torch.nn.Linear(64, 6), torch.nn.Tanh()).cuda()
with torch.no_grad():
    c = pol(torch.cat([q, v], dim=1).float()).double().contiguous()     # closed loop: the nominal action IS the policy's
out = e.new_outputs(N)
for mode in (0, 1):
    for _ in range(2):
        cp = e.perturbed_policy_actions(q, v, pol, 1e-6, mode)
        e.fd_jacobians(q, v, c, fd_mode=mode, ctrl_pert=cp, out=out, planned=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    cp = e.perturbed_policy_actions(q, v, pol, 1e-6, mode)
    ev[1].record()
    e.fd_jacobians(q, v, c, fd_mode=mode, ctrl_pert=cp, out=out, planned=True)
    ev[2].record(); torch.cuda.synchronize()
    print("mode", mode, "policy on perturbed states %.3f ms, linearisation %.3f ms" % (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])),
          "peak mem %.2f GB" % (torch.cuda.max_memory_allocated() / 2**30), flush=True)
