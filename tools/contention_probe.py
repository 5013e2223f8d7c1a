"""Does a wave run slower when every SIMD of the chip is busy?  One unit per wave (MBRL_DEBUG_SPREAD=1): 64 waves vs 1024
waves of the same per-wave work, for kernels of different code size."""
import os, sys
os.environ["MBRL_DEBUG_SPREAD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine, workloads
for env in ("SwimmerEnv", "HopperEnv", "HalfCheetahEnv", "Walker2dEnv"):
    e = engine.FdEngine(env, 10, "cuda:0")
    q0, v0, c0 = workloads.s2(e, 64, seed=0, horizon=20)
    for reps in (1, 4, 16, 64):     # the same 64 units, repeated: 64 .. 4096 one-lane waves
        q, v, c = (x.repeat(reps, 1).contiguous() for x in (q0, v0, c0))
        for _ in range(3): e.step(q, v, c)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20): e.step(q, v, c)
        b.record(); torch.cuda.synchronize()
        print("%-16s %5d one-lane waves: %.1f us per launch (10 mj_step)" % (env, 64 * reps, a.elapsed_time(b) / 20 * 1e3), flush=True)
