#!/usr/bin/env python3
"""Measurement script (not collected by pytest; lives under tests/ because its CPU-baseline leg uses the oracle).
Measurement sweep of SURVEY.md 8(d), on the GPU box: every env, forward (reference parity) and central
differences, N = 2^10 .. 2^18 for the headline env, counted flops per unit, share of constrained units, and the CPU
oracle on 1 thread / 16 threads.  Writes a markdown report (default gpurun_out/sweep.md)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import mbrl_amd
from mbrl_amd import workloads
from mbrl_amd.engine import FdEngine
from oracle import Oracle
import util

PEAK = 78.6e12
TW = C.CDLL(os.path.join(ROOT, "tests", "_build", "libhost_twin.so"))
TW.twin_count_flops_mt.restype = C.c_longlong
dp = C.POINTER(C.c_double)


def kernel_ms(eng, q, v, c, fd_mode, reps=7, planned=True):
    out = eng.fd_jacobians(q, v, c, fd_mode=fd_mode, planned=planned)
    eng.fd_jacobians(q, v, c, fd_mode=fd_mode, out=out, planned=planned)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); eng.fd_jacobians(q, v, c, fd_mode=fd_mode, out=out, planned=planned); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[reps // 2], out


def flops_per_unit(eng, eid, q, v, c, fd_mode, eps=1e-6):
    """Counted on the host build of the kernel's own simulator (FMA = 2, others = 1), same perturbation schedule."""
    n, nv, nu = q.shape[0], eng.nv, eng.nu
    npert = nu + 2 * nv
    R = 1 + (2 * npert if fd_mode else npert)
    Q, V, Cc = np.repeat(q, R, 0), np.repeat(v, R, 0), np.repeat(c, R, 0)
    for p in range(1, R):
        i = (p - 1) >> 1 if fd_mode else p - 1
        d = (-eps if (p - 1) & 1 else eps) if fd_mode else eps
        tgt = Cc if i < nu else (V if i < nu + nv else Q)
        col = i if i < nu else (i - nu if i < nu + nv else i - nu - nv)
        tgt[p::R, col] += d
    qn, vn, r = np.empty_like(Q), np.empty_like(V), np.empty(len(Q))
    st = np.zeros(len(Q), dtype=np.uint32)
    fl = TW.twin_count_flops_mt(eid, 0, C.c_int64(len(Q)), Q.ctypes.data_as(dp), V.ctypes.data_as(dp), Cc.ctypes.data_as(dp),
                                qn.ctypes.data_as(dp), vn.ctypes.data_as(dp), r.ctypes.data_as(dp),
                                st.ctypes.data_as(C.POINTER(C.c_uint32)), 16)
    return fl / n


def constrained_share(eng, eid, q, v, c):
    """share of units with an active-candidate constraint at t0, and mean Newton passes (Hessian builds) of those"""
    k, passes = 0, 0
    qa, fs, fc = np.zeros(eng.nv), np.zeros(eng.nv), np.zeros(eng.nv)
    it, na = C.c_int(), C.c_int()
    for i in range(len(q)):
        TW.twin_forward(eid, q[i].ctypes.data_as(dp), v[i].ctypes.data_as(dp), c[i].ctypes.data_as(dp), qa.ctypes.data_as(dp),
                        fs.ctypes.data_as(dp), fc.ctypes.data_as(dp), C.byref(it), C.byref(na))
        k += it.value > 0
        passes += it.value
    constrained_share.passes = passes / max(k, 1)
    return k / len(q)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "sweep.md")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    L = ["# Measurement sweep (one MI355X; S2 on-trajectory states; kernel time from HIP events, inputs resident in HBM)", "",
         "| env | mode | units | single launch ms | plan + planned ms | Jacobians/s (faster of the two) | counted flops/unit | flops per dynamics evaluation | fp64 TFLOP/s | frac of 78.6 | constrained at t0 | Newton passes per constrained evaluation | bad status |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    cpu = ["", "| env | CPU oracle 1 thread (Jacobians/s) | 16 threads | GPU / 1 thread |", "|---|---|---|---|"]
    for eid, env in enumerate(util.ENVS):
        eng = FdEngine(env)
        N = 65536
        q, v, c = workloads.s2(eng, N, seed=0)
        nsample = 256 if env == "Walker2dEnv" else 2048
        qh, vh, ch = (x[:nsample].cpu().numpy().copy() for x in (q, v, c))
        evals = {"InvertedPendulumEnv": 48, "InvertedDoublePendulumEnv": 128, "SwimmerEnv": 520, "HopperEnv": 320,
                 "Walker2dEnv": 2500, "HalfCheetahEnv": 250}[env]           # BASELINE.md section 2, forward mode
        share = constrained_share(eng, eid, *(x[:512].cpu().numpy().copy() for x in (q, v, c)))
        rate0 = None
        for fd_mode in (0, 1):
            ms1, o = kernel_ms(eng, q, v, c, fd_mode, planned=False)
            ms2, o = kernel_ms(eng, q, v, c, fd_mode, planned=True)
            ms = min(ms1, ms2)
            fpu = flops_per_unit(eng, eid, qh, vh, ch, fd_mode)
            rate = N / ms * 1e3
            rate0 = rate0 or rate
            ev = evals * (2 * (eng.n + eng.nu) + 1) / (eng.n + eng.nu + 1) if fd_mode else evals
            L.append("| %s | %s | %d | %.3f | %.3f | %.3e | %.3e | %.0f | %.2f | %.1f %% | %.0f %% | %.2f | %d |" %
                     (env[:-3], "central" if fd_mode else "forward (parity)", N, ms1, ms2, rate, fpu, fpu / ev,
                      rate * fpu / 1e12, 100 * rate * fpu / PEAK, 100 * share, constrained_share.passes,
                      int((o["status"] != 0).sum())))
            print(L[-1], flush=True)
        orc = Oracle(env)
        n1 = {"Walker2dEnv": 256, "HalfCheetahEnv": 4096, "HopperEnv": 4096}.get(env, 16384)
        qc, vc, cc = (x[:4 * n1].cpu().numpy().copy() for x in (q, v, c))
        orc.fd_jacobians(qc[:8], vc[:8], cc[:8])
        t0 = time.perf_counter(); orc.fd_jacobians(qc[:n1], vc[:n1], cc[:n1], nthreads=1); t1 = time.perf_counter()
        orc.fd_jacobians(qc, vc, cc, nthreads=16); t2 = time.perf_counter()
        cpu.append("| %s | %.0f | %.0f | %.0fx |" % (env[:-3], n1 / (t1 - t0), 4 * n1 / (t2 - t1), rate0 / (n1 / (t1 - t0))))
        print(cpu[-1], flush=True)
    L += cpu
    eng = FdEngine("HalfCheetahEnv")
    # S1 (reset-like) next to S2 for the headline env: BASELINE.md section 3
    L += ["", "| HalfCheetah set | units | plan + planned ms | Jacobians/s | counted flops/unit forward | central | flops per evaluation | constrained at t0 | Newton passes per constrained evaluation |",
          "|---|---|---|---|---|---|---|---|---|"]
    for name, (qs, vs, cs) in (("S1 reset-like", workloads.s1(eng, 163840, seed=0)), ("S2 on-trajectory", workloads.s2(eng, 163840, seed=0))):
        ms, _ = kernel_ms(eng, qs, vs, cs, 0, planned=True)
        qh, vh, ch = (x[:2048].cpu().numpy().copy() for x in (qs, vs, cs))
        f0, f1 = flops_per_unit(eng, 5, qh, vh, ch, 0), flops_per_unit(eng, 5, qh, vh, ch, 1)
        share = constrained_share(eng, 5, *(x[:2048].cpu().numpy().copy() for x in (qs, vs, cs)))
        L.append("| %s | 163840 | %.3f | %.3e | %.3e | %.3e | %.0f | %.0f %% | %.2f |" %
                 (name, ms, 163840 / ms * 1e3, f0, f1, f0 / 250, 100 * share, constrained_share.passes))
        print(L[-1], flush=True)
    L += ["", "| HalfCheetah forward, units per launch | single launch ms | plan + planned ms | Jacobians/s (faster) |", "|---|---|---|---|"]
    qa, va, ca = workloads.s2(eng, 1 << 18, seed=1)
    for k in range(8, 19, 2):
        n = 1 << k
        ms1, _ = kernel_ms(eng, qa[:n].contiguous(), va[:n].contiguous(), ca[:n].contiguous(), 0, planned=False)
        ms2, _ = kernel_ms(eng, qa[:n].contiguous(), va[:n].contiguous(), ca[:n].contiguous(), 0, planned=True)
        L.append("| 2^%d = %d | %.3f | %.3f | %.3e |" % (k, n, ms1, ms2, n / min(ms1, ms2) * 1e3))
        print(L[-1], flush=True)
    open(out_path, "w").write("\n".join(L) + "\n")


if __name__ == "__main__":
    main()
