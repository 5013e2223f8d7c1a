#!/usr/bin/env python3
"""Profiling aid: how unevenly the waves of ONE step launch finish (HalfCheetah, B episodes of on-trajectory states).
A step launch lasts as long as its slowest wave; a rollout of several env steps in one launch would last as long as the
slowest SUM over its steps.  Uses the -DMBRL_WAVE_TIMING build (python tools/wave_timing.py build)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "_build", "libmbrl_timing.so")
import numpy as np
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import lib, workloads
from mbrl_amd.engine import FdEngine

env = os.environ.get("ENV", "HalfCheetahEnv")
B, H = int(os.environ.get("B", 4096)), int(os.environ.get("H", 40))
lib.LIB_PATH = OUT
eng = FdEngine(env)
q, v, c = workloads.s1(eng, B, seed=0)
nblk = 2 * ((B + 127) // 128)
buf = torch.zeros((nblk, 4), dtype=torch.int64, device="cuda")
assert eng.L.mbrl_debug_set_wave_timing(C.c_void_p(buf.data_ptr())) == 0
g = torch.Generator(device="cpu").manual_seed(7)
per_step = []
for t in range(H):
    a = torch.clamp(0.5 * torch.randn((B, eng.nu), generator=g, dtype=torch.float64), -1, 1).cuda()
    buf.zero_()
    qn, vn, _, st = eng.step(q, v, a)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().astype(np.float64)
    comp = (s[:, 1] - s[:, 0])[s[:, 1] > 0]
    per_step.append(comp)
    q, v = qn, vn
W = np.array(per_step)                      # [H, waves] shader clocks per wave and step
print("%s B=%d: %d waves per launch" % (env, B, W.shape[1]))
print("per step: mean wave %.0f clocks, slowest wave %.0f (mean over steps); slowest/mean = %.2f" %
      (W.mean(), W.max(1).mean(), W.max(1).mean() / W.mean()))
for k in (1, 2, 4, 8, 40):
    sums = W[:H // k * k].reshape(H // k, k, -1).sum(1)       # each wave keeps its episodes: k steps in one launch
    print("  %2d env steps per launch: sum over launches of the slowest wave = %.0f clocks (%.2f of one-step launches)" %
          (k, sums.max(1).sum(), sums.max(1).sum() / W[:H // k * k].max(1).sum()))
