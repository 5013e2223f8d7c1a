"""Profiling driver (not a test): a few launches of the FD kernel on S2-like HalfCheetah states.
usage: python tools/prof_fd.py [env] [N] [launches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mbrl_amd
from mbrl_amd import workloads
from mbrl_amd.engine import FdEngine
env = sys.argv[1] if len(sys.argv) > 1 else "HalfCheetahEnv"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
L = int(sys.argv[3]) if len(sys.argv) > 3 else 3
e = FdEngine(env)
q, v, c = workloads.s2(e, N, seed=0)
out = e.fd_jacobians(q, v, c, planned=True); torch.cuda.synchronize()   # the bench schedule: plan + perturbed rollouts
t0 = time.time()
for _ in range(L): e.fd_jacobians(q, v, c, out=out, planned=True)
torch.cuda.synchronize(); dt = (time.time() - t0) / L
print(env, "N", N, "ms %.3f" % (dt * 1e3), "J/s %.3e" % (N / dt), "bad", int((out["status"] != 0).sum()), flush=True)
