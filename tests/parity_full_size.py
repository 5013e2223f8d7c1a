#!/usr/bin/env python3
"""Checker script (not collected by pytest; uses the oracle): the HIP linearisation of the FULL bench batch (HalfCheetah,
4096 x 40 = 163 840 S2 units, forward differences, the two-launch schedule bench.py times) against the oracle on every
unit -- the parity tests do this on hundreds of units, this run on all of them.  Prints the margins."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import workloads
from mbrl_amd.engine import FdEngine
from oracle import Oracle

env = sys.argv[1] if len(sys.argv) > 1 else "HalfCheetahEnv"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 163840
fd_mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # 0 forward (reference parity), 1 central
e, o = FdEngine(env), Oracle(env)
q, v, c = workloads.s2(e, N, seed=0, horizon=40)
out = {k: t.cpu().numpy() for k, t in e.fd_jacobians(q, v, c, fd_mode=fd_mode, planned=True).items()}
t0 = time.time()
ref = o.fd_jacobians(q.cpu().numpy(), v.cpu().numpy(), c.cpu().numpy(), fd_mode=fd_mode, nthreads=16)
print("%s, %s differences: %d units, oracle %.1f s on 16 threads" % (env, "central" if fd_mode else "forward", N, time.time() - t0), flush=True)
print("status words equal: %s; flagged units: %d" % (bool((out["status"] == ref["status"].astype(np.int32)).all()),
                                                      int((ref["status"] != 0).sum())))
ok = ref["status"] == 0
for k in ("s_next", "reward"):
    a, b = out[k][ok].reshape(ok.sum(), -1), ref[k][ok].reshape(ok.sum(), -1)
    print("%-7s max |diff| / (1 + max|ref| of the unit)  %.2e" % (k, (np.abs(a - b) / (1 + np.abs(b).max(1, keepdims=True))).max()))
for k in ("dsds", "dsda", "drds", "drda"):
    a, b = out[k][ok].reshape(ok.sum(), -1), ref[k][ok].reshape(ok.sum(), -1)
    err = (np.abs(a - b) / (1.0 + np.abs(b).max(axis=1, keepdims=True))).max(axis=1)
    print("%-7s max %.2e   99.9 %% of units below %.2e   units above 5e-7: %d" % (k, err.max(), np.quantile(err, 0.999), int((err > 5e-7).sum())))
