#!/usr/bin/env python3
"""A/B timing of library variants on the bench workload (HalfCheetah S2, forward FD).

CPU box:  python tools/ab_bench.py build NAME [extra hipcc flags...]   -> tests/_build/variants/NAME.so
GPU box:  python tools/ab_bench.py run [NAME ...]                       (default: every built variant)
Each variant runs in its own child process (one library instance per process)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tests", "_build", "variants")


def build(name, flags):
    import mbrl_amd
    from mbrl_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    b.compile_lib(os.path.join(VDIR, name + ".so"), flags)


def child(path):
    import torch
    import mbrl_amd
    from mbrl_amd import lib, workloads
    from mbrl_amd.engine import FdEngine
    if path != "default":
        lib.LIB_PATH = path
    envs = os.environ.get("ENVS", "HalfCheetahEnv").split(",")
    for env in envs:
        eng = FdEngine(env)
        N = int(os.environ.get("UNITS", 163840))
        q, v, c = workloads.s2(eng, N, seed=0)
        out = eng.fd_jacobians(q, v, c)
        for _ in range(2):
            eng.fd_jacobians(q, v, c, out=out)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for s, e in ev:
            s.record(); eng.fd_jacobians(q, v, c, out=out); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in ev)[len(ev) // 2]
        chk = float(out["dsds"].double().abs().sum().item())
        print("%-28s %-26s %9.3f ms  %8.3f M J/s  bad %d  checksum %.12e" %
              (os.path.basename(path), env, ms, N / ms / 1e3, int((out["status"] != 0).sum()), chk), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        names = sys.argv[2:] or ["default"] + sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(VDIR, "*.so")))
        for n in names:
            p = "default" if n == "default" else os.path.join(VDIR, n + ".so")
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), "child", p], timeout=300)
            if rc != 0:
                print("variant %s failed rc=%d; stopping" % (n, rc)); sys.exit(rc)
