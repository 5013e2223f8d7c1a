#!/usr/bin/env python3
"""bench.py -- FD dynamics Jacobians/sec (HalfCheetah, batch x horizon) on N MI355X GPUs.

One "step" = one linearisation of the whole batch through the FD-Jacobian hot path (reference parity mode: forward
differences, eps 1e-6, 1+nu+nv+nq = 25 rollouts x 10 mj_step per unit) over 4096 x 40 = 163 840 synthetic on-trajectory
(state, action) units per GPU, inputs resident in HBM.  Units are independent, so ranks shard them with no data-path
collective (weak scaling).

`python bench.py --gpus N` starts its own N ranks (fresh child processes, before anything touches the GPU) when it was
not started by a launcher; under `python -m torch.distributed.run ... bench.py --gpus N` it takes RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.

Prints ONE JSON line (rank 0).  See DESIGN.md for the roofline accounting.
"""
import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X spec: 256 CU x 4 SIMD x 16 fp64 FMA lanes/clk x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
ENV = "HalfCheetahEnv"
BATCH, HORIZON = 4096, 40
FLOPS_PER_UNIT_RECORDED = 0.61e6  # fallback; re-counted live when the host twin is present
FLOPS_PER_UNIT_ROUND2 = 0.71e6    # what round 2's formulation spent on the same Jacobians (profiles/r02_v15_bench.json)
FLOP_SAMPLE_UNITS = 4096


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--units", type=int, default=BATCH * HORIZON)
    ap.add_argument("--fd-mode", type=int, default=0)
    ap.add_argument("--cpu-units", type=int, default=16384)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the central-difference figures in `extra`")
    ap.add_argument("--single-launch", action="store_true", help="time mbrl_fd_jacobians instead of plan + planned")
    ap.add_argument("--rank-timeout", type=float, default=None,
                    help="--gpus N self-launcher: stop every rank after this many seconds (default: MBRL_RANK_TIMEOUT or none)")
    ap.add_argument("--serial-plan", action="store_true",
                    help="plan launch and perturbed rollouts of a step strictly one after the other on one stream "
                         "(default: the plan of step k+1 overlaps the perturbed rollouts of step k on a second stream)")
    return ap.parse_args()


def launch_ranks(n, timeout=None):
    """Start n ranks of this script as fresh child processes (this process has not touched the GPU and never will:
    nothing is exec'ed after HIP initialisation) and supervise them (mbrl_amd/launch.py): the first rank that fails ends
    the job -- its siblings are stopped instead of waiting in a collective -- and its exit code is returned
    (128 + signal for a signal death).  Rank 0's stdout is ours."""
    import mbrl_amd  # noqa: F401  (import shim; no torch, no GPU)
    from mbrl_amd import launch
    return launch.launch(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], timeout=timeout)


def recorded(pattern):
    """Newest committed profiles/rNN_<pattern>.json (rocprofv3 PMC passes of this command, one counter group per run),
    or None."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*" + pattern + ".json")))
    return (json.load(open(files[-1])), os.path.basename(files[-1])) if files else (None, None)


def count_flops_per_unit(q, v, c, frame_skip, fd_mode=0, eps=1e-6):
    """Algorithmic flops of one unit (all its rollouts) counted by running the kernel's own per-lane simulator template
    on the host with a counting scalar (tests/host_twin.cpp: FMA = 2, add / mul / div / sqrt / transcendental = 1;
    operations with a compile-time 0 or +-1 operand, which the device compiler folds away, are not charged)."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "_build", "libhost_twin.so")
    if not os.path.exists(path):
        return FLOPS_PER_UNIT_RECORDED, "recorded"
    L = C.CDLL(path)
    L.twin_count_flops_mt.restype = C.c_longlong
    dp = C.POINTER(C.c_double)
    n, nv, nu = q.shape[0], q.shape[1], c.shape[1]
    npert = nu + 2 * nv
    R = 1 + (2 * npert if fd_mode else npert)
    Q, V, Cc = np.repeat(q, R, 0), np.repeat(v, R, 0), np.repeat(c, R, 0)
    for p in range(1, R):   # same perturbation schedule as the kernel
        i = (p - 1) >> 1 if fd_mode else p - 1
        d = (-eps if (p - 1) & 1 else eps) if fd_mode else eps
        tgt = Cc if i < nu else (V if i < nu + nv else Q)
        col = i if i < nu else (i - nu if i < nu + nv else i - nu - nv)
        tgt[p::R, col] += d
    qn, vn, r = np.empty_like(Q), np.empty_like(V), np.empty(len(Q))
    st = np.zeros(len(Q), dtype=np.uint32)
    fl = L.twin_count_flops_mt(5, frame_skip, C.c_int64(len(Q)), Q.ctypes.data_as(dp), V.ctypes.data_as(dp),
                               Cc.ctypes.data_as(dp), qn.ctypes.data_as(dp), vn.ctypes.data_as(dp),
                               r.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_uint32)),
                               min(16, os.cpu_count() or 1))
    return fl / n, "counted on %d units" % n


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
    # all-core leg: every core this process may run on (the GPU box gives a one-GPU job a 16-CPU share of a much
    # larger host; os.cpu_count() reports the whole machine and is printed next to it)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores) or cores)))
    m = min(len(q), n * 4)
    t2 = time.perf_counter()
    o.fd_jacobians(q[:m], v[:m], c[:m], nthreads=cores)
    t3 = time.perf_counter()
    return {"value": n / (t1 - t0), "unit": "Jacobians/s", "cores": 1, "kind": "port",
            "sample": "%d HalfCheetah S2 units, oracle/mjc_oracle.c single thread (%.1f s)" % (n, t1 - t0),
            "all_core_value": m / (t3 - t2), "all_core_threads": cores, "host_cpu_count": os.cpu_count(),
            "all_core_note": "threads = the cores this job may use (affinity / OMP_NUM_THREADS share of the box), "
                             "host_cpu_count = os.cpu_count() of the whole machine"}


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, a.rank_timeout))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MBRL_FORCE_DIST=1: go through the process group with ONE rank too -- init_process_group("nccl"), the barriers, the
    # MAX reduce of the timing -- so that the RCCL branch runs on a one-GPU box exactly as it will on eight
    force = os.environ.get("MBRL_FORCE_DIST", "") not in ("", "0")
    dist = world > 1 or force
    # MBRL_DIST_BACKEND=gloo: rehearsal with more ranks than GPUs (ranks share GPUs; RCCL needs a GPU per rank)
    backend = os.environ.get("MBRL_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if dist:
        import torch.distributed as td
        import mbrl_amd  # noqa: F401
        from mbrl_amd import launch
        if "MASTER_ADDR" not in os.environ:      # forced single rank without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(launch.free_port()), RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=launch.init_timeout())
        else:
            td.init_process_group(backend, timeout=launch.init_timeout())
        for _ in range(3):       # the first collectives set up channels lazily: keep that out of the timed region's barrier
            td.barrier()

    import mbrl_amd  # noqa: F401
    from mbrl_amd import workloads
    from mbrl_amd.engine import FdEngine

    eng = FdEngine(ENV, device="cuda:%d" % local)
    N = a.units
    q, v, c = workloads.s2(eng, N, seed=rank, horizon=HORIZON)   # each rank owns its own shard of units
    main_stream = torch.cuda.current_stream()

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    def timed(fd_mode, steps, warmup, serial=False):
        """K linearisations of the batch.  One step = mbrl_fd_plan (nominal rollouts + unit order) and
        mbrl_fd_jacobians_planned (the perturbed rollouts: the dominant kernel); --single-launch: mbrl_fd_jacobians.
        Default schedule: two plan buffers; the plan of step k+1 is launched on a second stream while the perturbed
        rollouts of step k run, each planned launch waits for its own plan (events), every step does all of its work
        inside the timed region.  Returns wall seconds, mean plan ms, mean kernel ms (HIP events on the stream the
        kernel runs on), outputs."""
        out = eng.new_outputs(N)
        pipelined = not (a.single_launch or a.serial_plan or serial)
        plans = [eng.new_plan(N) for _ in range(2 if pipelined else 1)]
        side = eng.side_stream() if pipelined else main_stream
        nom = [dict(s_next=out["s_next"], reward=out["reward"], status=eng._new(N, dtype=torch.int32)) for _ in plans]

        def plan_on(stream, k):
            with torch.cuda.stream(stream):
                eng.fd_plan(q, v, c, out=nom[k % len(plans)], plan=plans[k % len(plans)])
                e = torch.cuda.Event()
                e.record(stream)
            return e

        def run(K, ev):
            if a.single_launch:
                for k in range(K):
                    if ev: ev[k][1].record(main_stream)
                    eng.fd_jacobians(q, v, c, fd_mode=fd_mode, out=out, planned=False)
                    if ev: ev[k][2].record(main_stream)
                return
            ready = plan_on(side, 0) if pipelined else None
            for k in range(K):
                if pipelined:
                    main_stream.wait_event(ready)
                    if k + 1 < K:
                        # plan k+1 may overwrite its buffer only after planned launch k-1 (same buffer) has finished
                        side.wait_stream(main_stream)
                        nxt = plan_on(side, k + 1)
                else:
                    if ev: ev[k][0].record(main_stream)
                    plan_on(main_stream, k)
                if ev: ev[k][1].record(main_stream)
                o = dict(out, status=nom[k % len(plans)]["status"])
                eng.fd_planned(q, v, c, o, fd_mode=fd_mode, plan=plans[k % len(plans)])
                if ev: ev[k][2].record(main_stream)
                if pipelined and k + 1 < K:
                    ready = nxt
            out["status"] = nom[(K - 1) % len(plans)]["status"]

        run(max(warmup, 1) if warmup else 0, None)
        ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        run(steps, ev)
        barrier()
        elapsed = time.perf_counter() - t0
        kern_ms = float(np.mean([m[1].elapsed_time(m[2]) for m in ev]))
        plan_ms = 0.0 if (a.single_launch or pipelined) else float(np.mean([m[0].elapsed_time(m[1]) for m in ev]))
        if dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            td.all_reduce(t, op=td.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, plan_ms, kern_ms, out, pipelined

    elapsed, plan_ms, kern_ms, out, pipelined = timed(a.fd_mode, a.steps, a.warmup)
    bad = int((out["status"] != 0).sum().item())
    extra = {}
    if not a.no_extra and a.fd_mode == 0:
        # north_star words the metric with central differences: the same batch, same schedule, driver-visible
        ks = max(3, min(a.steps, 10))
        el2, p2, k2, _, _ = timed(1, ks, 1)
        extra["central"] = {"value": world * N * ks / el2, "unit": "Jacobians/s", "steps": ks, "ms_per_step": el2 / ks * 1e3,
                            "kernel_ms": k2, "plan_ms": p2, "rollouts_per_unit": 2 * (eng.n + eng.nu) + 1}
    if not a.no_extra and pipelined:
        # the same step with its two launches strictly one after the other on one stream: the dominant kernel alone
        # on the chip (its un-overlapped duration) and the plan launch's own cost
        ks = max(3, min(a.steps, 10))
        el3, p3, k3, _, _ = timed(a.fd_mode, ks, 1, serial=True)
        extra["serial_plan"] = {"value": world * N * ks / el3, "unit": "Jacobians/s", "steps": ks,
                                "ms_per_step": el3 / ks * 1e3, "kernel_ms_alone": k3, "plan_ms": p3}

    if rank == 0:
        n, nu = eng.n, eng.nu
        sel = torch.arange(0, N, max(1, N // FLOP_SAMPLE_UNITS), device=q.device)[:FLOP_SAMPLE_UNITS]
        qh, vh, ch = q[sel].cpu().numpy(), v[sel].cpu().numpy(), c[sel].cpu().numpy()
        fpu, how = count_flops_per_unit(qh, vh, ch, 10, a.fd_mode)
        R = 2 * (n + nu) + 1 if a.fd_mode == 1 else n + nu + 1
        fpu_kernel = fpu if a.single_launch else fpu * (R - 1) / R   # the planned kernel runs the perturbed ones only
        bytes_unit = (n + nu) * 8 + (n * n + n * nu + 2 * n + nu + 1) * 8 + 4
        flops_alg = fpu_kernel * N
        # executed lane-flops of the same kernel from the committed PMC pass: 64 x (2 FMA + ADD + MUL + TRANS) fp64
        # VALU instructions -- an upper bound on useful work (an instruction with a partial EXEC mask counts 64 lanes)
        pmc, pmc_file = recorded("pmc_flops")
        flops_exec = None
        if pmc and pmc.get("fd_mode", 0) == a.fd_mode and not a.single_launch:
            flops_exec = pmc["lane_flops_per_launch"] * N / pmc["units"]
        flops_used = min(flops_alg, flops_exec) if flops_exec else flops_alg
        tflops = flops_used / (kern_ms * 1e-3) / 1e12
        gbs = bytes_unit * N / (kern_ms * 1e-3) / 1e9
        traffic, traffic_file = recorded("traffic")
        line = {
            "metric": "FD dynamics Jacobians/sec (HalfCheetah, batch x horizon)",
            "value": world * N * a.steps / elapsed, "unit": "Jacobians/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "HalfCheetah FD Jacobians, %s differences eps=1e-6, %d units/GPU (batch %d x horizon %d), "
                                   "S2 on-trajectory states" % ("central" if a.fd_mode else "forward (reference parity)",
                                                                N, BATCH, HORIZON),
                       "launches": "mbrl_fd_jacobians" if a.single_launch else
                                   "mbrl_fd_plan + mbrl_fd_jacobians_planned" +
                                   (" (plan of step k+1 on a second stream under the perturbed rollouts of step k)"
                                    if pipelined else ""),
                       "env": ENV, "units_per_gpu": N, "frame_skip": 10, "fd_mode": a.fd_mode,
                       "side_stream_candidates_rejected": getattr(eng, "side_stream_rejected", None),
                       "parallelism": "units sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {"bound": "valu_fp64", "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                         "traffic": traffic["hbm_bytes_per_launch"] * N / traffic["units"] if traffic else None,
                         "traffic_note": "HBM bytes per launch from %s (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 "
                                         "correction); above the %d algorithmic bytes because inputs are gathered per lane -- "
                                         "harmless here, the kernel is VALU-bound" % (traffic_file, bytes_unit * N)
                                         if traffic else None,
                         "kernel": "rollout_kernel<MdlHalfCheetah, false, false>" +
                                   ("" if a.single_launch else " launched by mbrl_fd_jacobians_planned"),
                         "kernel_ms": kern_ms, "plan_ms": plan_ms,
                         "flops_per_launch_algorithmic": flops_alg, "flops_per_launch_executed_pmc": flops_exec,
                         "flops_executed_source": pmc_file if flops_exec else None,
                         "frac_basis": "min(algorithmic, executed) / kernel_ms / peak; kernel_ms is the kernel's duration "
                                       "inside the timed region (with the next step's plan launch sharing the chip when "
                                       "pipelined; extra.serial_plan.kernel_ms_alone is the un-overlapped duration)",
                         "flops_per_unit": fpu, "flops_per_unit_in_kernel": fpu_kernel, "flops_source": how,
                         "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                 "bytes_per_unit": bytes_unit},
                         "note": "path is fp64-VALU bound (arithmetic intensity ~%d flop/B); neither HBM nor MFMA binds"
                                 % int(fpu / bytes_unit)},
            "units_with_bad_status": bad,
        }
        if "serial_plan" in extra:
            alone = extra["serial_plan"]["kernel_ms_alone"]
            extra["serial_plan"]["roofline_frac_alone"] = flops_used / (alone * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS
            if a.fd_mode == 0 and not a.single_launch:
                # the same launch priced with the flop count of round 2's formulation (the shared joint-space inertia of
                # round 3 removed a fifth of the counted work together with the time): for comparison across rounds only
                extra["serial_plan"]["roofline_frac_alone_at_round2_flop_count"] = \
                    FLOPS_PER_UNIT_ROUND2 * (R - 1) / R * N / (alone * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS
        if extra:
            line["extra"] = extra
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
