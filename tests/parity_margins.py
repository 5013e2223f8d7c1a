#!/usr/bin/env python3
"""Checker script (not collected by pytest; lives under tests/ because it uses the oracle).
Report how far the HIP path is from the oracle (same measure as tests/test_gpu_parity.py) -- margins, not a test."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import mbrl_amd
from mbrl_amd.engine import FdEngine
from oracle import Oracle
import util
for env in util.ENVS:
    o, e = Oracle(env), FdEngine(env)
    for fd_mode in (0, 1):
        q, v, c = util.on_trajectory(o, env, 300, seed=1)
        ref = o.fd_jacobians(q, v, c, fd_mode=fd_mode, nthreads=8)
        out = {k: t.cpu().numpy() for k, t in e.fd_jacobians(q, v, c, fd_mode=fd_mode).items()}
        ok = ref["status"] == 0
        line = "%-26s mode %d  s' %.1e" % (env, fd_mode, util.relerr(out["s_next"][ok], ref["s_next"][ok]))
        for k in ("dsds", "dsda", "drds", "drda"):
            a, b = out[k][ok].reshape(ok.sum(), -1), ref[k][ok].reshape(ok.sum(), -1)
            line += "  %s %.1e" % (k, (np.abs(a - b) / (1.0 + np.abs(b).max(axis=1, keepdims=True))).max())
        print(line + "   (bound 2e-6; status equal: %s)" % bool((out["status"] == ref["status"].astype(np.int32)).all()), flush=True)
