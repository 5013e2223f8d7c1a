#!/usr/bin/env python3
"""Profiling aid: per-wave shader-clock stamps of the FD rollout kernel (HalfCheetah bench workload).

Builds a -DMBRL_WAVE_TIMING variant of the library into tests/_build/ (on the CPU box:
`python tools/wave_timing.py build`) and, on the GPU box (`python tools/wave_timing.py`), reports how
long the waves of a workgroup compute vs. wait at the post-rollout barrier."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "_build", "libmbrl_timing.so")


def build():
    import mbrl_amd
    from mbrl_amd import build as b
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    b.compile_lib(OUT, ["-DMBRL_WAVE_TIMING"])


def main():
    import numpy as np
    import torch
    import mbrl_amd
    from mbrl_amd import lib, workloads
    from mbrl_amd.engine import FdEngine
    eng0 = FdEngine("HalfCheetahEnv")
    N = int(os.environ.get("UNITS", 163840))
    q, v, c = workloads.s2(eng0, N, seed=0)
    lib._lib = None
    lib.LIB_PATH = OUT
    eng = FdEngine("HalfCheetahEnv")
    upb = int(os.environ.get("UPB", 5)); wpb = 1; nblk = 2 * ((N + upb - 1) // upb)
    buf = torch.zeros((nblk * wpb, 4), dtype=torch.int64, device="cuda")
    assert eng.L.mbrl_debug_set_wave_timing(C.c_void_p(buf.data_ptr())) == 0
    out = eng.fd_jacobians(q, v, c)
    eng.fd_jacobians(q, v, c, out=out)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nblk, wpb, 4).astype(np.float64)
    comp = t[:, :, 1] - t[:, :, 0]          # rollout compute per wave
    wait = t[:, :, 2] - t[:, :, 1]          # barrier wait per wave
    tail = t[:, :, 3] - t[:, :, 2]          # Jacobian assembly + stores
    wg = t[:, :, 3].max(1) - t[:, :, 0].min(1)
    print("clock ticks (s_memtime): per-wave compute mean %.0f  barrier wait mean %.0f  assembly mean %.0f" %
          (comp.mean(), wait.mean(), tail.mean()))
    print("workgroup span mean %.0f ; mean-over-waves compute / workgroup span = %.3f" % (wg.mean(), comp.mean() / wg.mean()))
    print("max-wave compute / mean-wave compute per workgroup: %.3f" % (comp.max(1) / comp.mean(1)).mean())
    print("compute percentiles (5,25,50,75,95):", np.percentile(comp, [5, 25, 50, 75, 95]).round(0))


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else main()
