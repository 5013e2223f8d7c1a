"""Fixed vs per-mj_step cost of one step launch (N = 4096): frame_skip sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine, workloads
e0 = engine.FdEngine("HalfCheetahEnv", 0, "cuda:0")
for N in (64, 4096):
    q, v, c = workloads.s2(e0, N, seed=0, horizon=40)
    for fs in (1, 2, 5, 10, 20, 40):
        e = engine.FdEngine("HalfCheetahEnv", fs, "cuda:0")
        for _ in range(3): e.step(q, v, c)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20): e.step(q, v, c)
        b.record(); torch.cuda.synchronize()
        print("N", N, "frame_skip", fs, "%.1f us per launch" % (a.elapsed_time(b) / 20 * 1e3), flush=True)
