#!/usr/bin/env python3
"""bench.py -- FD dynamics Jacobians/sec (HalfCheetah, batch x horizon) on N MI355X GPUs.

One "step" = one launch of the FD-Jacobian hot path (mbrl_fd_jacobians, reference parity mode:
forward differences, eps 1e-6, 1+nu+nv+nq = 25 rollouts x 10 mj_step per unit) over one batch of
4096 x 40 = 163 840 synthetic on-trajectory (state, action) units per GPU, inputs resident in HBM.
Units are independent, so ranks shard them with no data-path collective (weak scaling).

Prints ONE JSON line (rank 0).  See DESIGN.md for the roofline accounting.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X spec: 256 CU x 4 SIMD x 16 fp64 FMA lanes/clk x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
ENV = "HalfCheetahEnv"
BATCH, HORIZON = 4096, 40
FLOPS_PER_UNIT_RECORDED = 1.0e6  # fallback; re-counted live when the host twin is present


def recorded_traffic(n_units):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_traffic.json; FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs, KB units, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950), scaled to this launch's unit count.  None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    return t["hbm_bytes_per_launch"] * n_units / t["units"]


def count_flops_per_unit(q, v, c, frame_skip, eps=1e-6):
    """Algorithmic flops of one unit (25 rollouts) counted by running the kernel's own per-lane
    simulator template on the host with a counting scalar (FMA = 2, add/mul/div/sqrt/transcendental = 1)."""
    path = os.path.join(ROOT, "tests", "_build", "libhost_twin.so")
    if not os.path.exists(path):
        return FLOPS_PER_UNIT_RECORDED, "recorded"
    L = C.CDLL(path)
    L.twin_count_flops.restype = C.c_longlong
    dp = C.POINTER(C.c_double)
    n, nv, nu = q.shape[0], q.shape[1], c.shape[1]
    R = 1 + nu + 2 * nv
    Q, V, Cc = np.repeat(q, R, 0), np.repeat(v, R, 0), np.repeat(c, R, 0)
    for p in range(1, R):   # same perturbation schedule as the kernel
        i = p - 1
        tgt = Cc if i < nu else (V if i < nu + nv else Q)
        col = i if i < nu else (i - nu if i < nu + nv else i - nu - nv)
        tgt[p::R, col] += eps
    qn, vn, r = np.empty_like(Q), np.empty_like(V), np.empty(len(Q))
    st = np.zeros(len(Q), dtype=np.uint32)
    fl = L.twin_count_flops(5, frame_skip, C.c_int64(len(Q)), Q.ctypes.data_as(dp), V.ctypes.data_as(dp),
                            Cc.ctypes.data_as(dp), qn.ctypes.data_as(dp), vn.ctypes.data_as(dp),
                            r.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_uint32)))
    return fl / n, "counted"


def cpu_baseline(q, v, c, budget_units):
    """The CPU oracle (oracle/, a MuJoCo-equivalent restatement -- NOT MuJoCo) on the same workload,
    single thread like the reference, on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle
    o = Oracle(ENV)
    n = min(budget_units, len(q))
    o.fd_jacobians(q[:8], v[:8], c[:8])
    t0 = time.perf_counter()
    o.fd_jacobians(q[:n], v[:n], c[:n], nthreads=1)
    t1 = time.perf_counter()
    cores = min(16, os.cpu_count() or 1)
    m = min(len(q), n * 4)
    t2 = time.perf_counter()
    o.fd_jacobians(q[:m], v[:m], c[:m], nthreads=cores)
    t3 = time.perf_counter()
    return {"value": n / (t1 - t0), "unit": "Jacobians/s", "cores": 1, "kind": "port",
            "sample": "%d HalfCheetah S2 units, oracle/mjc_oracle.c single thread (%.1f s)" % (n, t1 - t0),
            "all_core_value": m / (t3 - t2), "all_core_threads": cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--units", type=int, default=BATCH * HORIZON)
    ap.add_argument("--fd-mode", type=int, default=0)
    ap.add_argument("--cpu-units", type=int, default=16384)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--single-launch", action="store_true", help="time mbrl_fd_jacobians instead of plan + planned")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    torch.cuda.set_device(local)
    if dist:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=torch.device("cuda", local))

    import mbrl_amd
    from mbrl_amd import workloads
    from mbrl_amd.engine import FdEngine

    eng = FdEngine(ENV, device="cuda:%d" % local)
    N = a.units
    q, v, c = workloads.s2(eng, N, seed=rank, horizon=HORIZON)   # each rank owns its own shard of units
    out = eng.fd_jacobians(q, v, c, fd_mode=a.fd_mode, planned=not a.single_launch)   # only the timed schedule ever runs
    torch.cuda.synchronize()

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    # One step = the whole linearisation of the batch: mbrl_fd_plan (nominal rollouts + unit order) and
    # mbrl_fd_jacobians_planned (the perturbed rollouts: the dominant kernel).  --single-launch times mbrl_fd_jacobians.
    def step(marks=None):
        if a.single_launch:
            eng.fd_jacobians(q, v, c, fd_mode=a.fd_mode, out=out, planned=False)
        else:
            eng.fd_plan(q, v, c, out=out)
            if marks:
                marks[1].record()
            eng.fd_planned(q, v, c, out, fd_mode=a.fd_mode)

    for _ in range(a.warmup):
        step()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(a.steps)]
    barrier()
    t0 = time.perf_counter()
    for m in ev:
        m[0].record()
        step(m)
        m[2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    # HIP events on the launch stream
    if a.single_launch:
        plan_ms, kern_ms = 0.0, float(np.mean([m[0].elapsed_time(m[2]) for m in ev]))
    else:
        plan_ms = float(np.mean([m[0].elapsed_time(m[1]) for m in ev]))
        kern_ms = float(np.mean([m[1].elapsed_time(m[2]) for m in ev]))
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
    bad = int((out["status"] != 0).sum().item())

    if rank == 0:
        n, nu = eng.n, eng.nu
        qh, vh, ch = q[:64].cpu().numpy(), v[:64].cpu().numpy(), c[:64].cpu().numpy()
        fpu, how = count_flops_per_unit(qh, vh, ch, 10)
        R = 2 * (n + nu) + 1 if a.fd_mode == 1 else n + nu + 1
        fpu *= R / (n + nu + 1)                      # flops of all R rollouts of a unit
        fpu_kernel = fpu if a.single_launch else fpu * (R - 1) / R   # the planned kernel runs the perturbed ones only
        bytes_unit = (n + nu) * 8 + (n * n + n * nu + 2 * n + nu + 1) * 8 + 4
        tflops = fpu_kernel * N / (kern_ms * 1e-3) / 1e12
        gbs = bytes_unit * N / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "FD dynamics Jacobians/sec (HalfCheetah, batch x horizon)",
            "value": world * N * a.steps / elapsed, "unit": "Jacobians/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "HalfCheetah FD Jacobians, %s differences eps=1e-6, %d units/GPU (batch %d x horizon %d), "
                                   "S2 on-trajectory states" % ("central" if a.fd_mode else "forward (reference parity)",
                                                                N, BATCH, HORIZON),
                       "launches": "mbrl_fd_jacobians" if a.single_launch else "mbrl_fd_plan + mbrl_fd_jacobians_planned",
                       "env": ENV, "units_per_gpu": N, "frame_skip": 10, "fd_mode": a.fd_mode,
                       "parallelism": "units sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {"bound": "valu_fp64", "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / FP64_VECTOR_PEAK_TFLOPS, "traffic": recorded_traffic(N),
                         "kernel": "rollout_kernel<MdlHalfCheetah, false>" +
                                   ("" if a.single_launch else " launched by mbrl_fd_jacobians_planned"),
                         "kernel_ms": kern_ms, "plan_ms": plan_ms,
                         "flops_per_unit": fpu, "flops_per_unit_in_kernel": fpu_kernel, "flops_source": how,
                         "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                 "bytes_per_unit": bytes_unit},
                         "note": "path is fp64-VALU bound (arithmetic intensity ~%d flop/B); neither HBM nor MFMA binds"
                                 % int(fpu / bytes_unit)},
            "units_with_bad_status": bad,
        }
        if not a.no_cpu:
            cb = cpu_baseline(q[:4 * a.cpu_units].cpu().numpy(), v[:4 * a.cpu_units].cpu().numpy(),
                              c[:4 * a.cpu_units].cpu().numpy(), a.cpu_units)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_1thread"] = line["value"] / cb["value"]
        print(json.dumps(line), flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
