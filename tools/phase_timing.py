#!/usr/bin/env python3
"""Profiling aid: shader-clock totals per solver phase of the FD rollout kernel (bench workload).
CPU box: python tools/phase_timing.py build ; GPU box: python tools/phase_timing.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "_build", "libmbrl_phase.so")
NAMES = {0: "sweep + smooth forces", 1: "unconstrained solve (crba, factor, solve)", 2: "newton pass (residuals, J^T f, Hessian)",
         3: "factor + solve + norms", 4: "M s for the line search", 5: "line search", 6: "solver exit", 7: "euler (crba, factor, solve, integrate)",
         8: "loop glue", 9: "reward + exit", 10: "(two calibration marks)"}


def build():
    import mbrl_amd
    from mbrl_amd import build as b
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    b.compile_lib(OUT, ["-DMBRL_PHASE_TIMING", "-DMBRL_ONLY_ENV=%d" % int(os.environ.get("ONLY_ENV", "5"))] + sys.argv[2:])


def main():
    import torch
    os.environ["MBRL_LIB"] = OUT
    import mbrl_amd
    from mbrl_amd import workloads
    from mbrl_amd.engine import FdEngine
    eng = FdEngine(os.environ.get("ENV", "HalfCheetahEnv"))
    N = int(os.environ.get("UNITS", 163840))
    q, v, c = workloads.s2(eng, N, seed=0)
    out = eng.fd_jacobians(q, v, c)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    assert eng.L.mbrl_debug_phase_read(buf, 1) == 0
    eng.fd_jacobians(q, v, c, out=out)
    torch.cuda.synchronize()
    assert eng.L.mbrl_debug_phase_read(buf, 0) == 0
    tot = float(sum(buf[:16]))
    mark = buf[10] / max(buf[26], 1)          # cost of one mark (phase 10 = two marks back to back)
    for k in range(16):
        if buf[k]:
            n = buf[16 + k]
            print("%-48s %6.2f %%   %9d times   %8.1f clocks each (mark cost %.0f not subtracted)" %
                  (NAMES.get(k, "phase %d" % k), 100.0 * buf[k] / tot, n, buf[k] / max(n, 1), mark))


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else main()
