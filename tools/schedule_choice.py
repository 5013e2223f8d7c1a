"""Which schedule FdEngine picks (single launch vs plan + planned) per env / mode / batch size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine, workloads
for env in ("InvertedPendulumEnv", "SwimmerEnv", "HopperEnv", "HalfCheetahEnv"):
    e = engine.FdEngine(env)
    for N in (4096, 32768, 163840):
        q, v, c = workloads.s2(e, N, seed=0)
        for mode in (0, 1):
            e.fd_jacobians(q, v, c, fd_mode=mode)
            print(env, "N", N, "central" if mode else "forward", "->", {k: ("planned" if p else "single") for k, p in e._schedule.items() if k[0] == mode and k[3] == N.bit_length()}, flush=True)
