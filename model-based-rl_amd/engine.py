"""Torch-facing wrapper of the C ABI: device tensors in, device tensors out.

PyTorch is plumbing here (HBM allocation, streams); all arithmetic happens in the HIP
kernels behind include/mbrl.h.
"""
import ctypes as C
import os

import torch

from . import lib, modelspec


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _runs_beside(engine, main, cand):
    """Does work on stream `cand` actually run WHILE a long launch on `main` is in flight?  Two HIP streams of one process
    can share a hardware queue (GPU_MAX_HW_QUEUES is 4 on the box, and everything that creates streams -- torch's pools,
    an RCCL communicator -- takes slots).  A queue hands its packets to the dispatcher in order, and a grid that is much
    larger than the chip holds its queue until its last workgroup has been placed: a launch queued behind it on the same
    queue starts when the big one is all but finished, whatever stream it came from.  HIP has no query for the mapping,
    so it is measured: a spin grid of 4 096 workgroups (two fit a CU: eight rounds of 50 us) on `main`, one short
    workgroup on `cand`; side by side they take the big grid's time, in one queue the sum."""
    L, h = engine.L, engine.h
    sm, sc = C.c_void_p(main.cuda_stream), C.c_void_p(cand.cuda_stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    big = lambda: lib.check(L.mbrl_stream_probe_spin(h, 4096, 50, 60 * 1024, sm))
    big()                                                    # warm-up (code load)
    lib.check(L.mbrl_stream_probe_spin(h, 1, 10, 0, sc))
    main.synchronize(); cand.synchronize()
    ev[0].record(main); big(); ev[1].record(main)
    ev[1].synchronize()
    alone = ev[0].elapsed_time(ev[1])
    ev[0].record(main)
    big()
    lib.check(L.mbrl_stream_probe_spin(h, 1, 200, 0, sc))    # 0.2 ms, queued while the big grid is being placed
    main.wait_stream(cand)
    ev[2].record(main)
    ev[2].synchronize()
    return ev[0].elapsed_time(ev[2]) < alone + 0.1


def concurrent_stream(engine, tries=8):
    """A stream of the engine's device that overlaps with the CURRENT stream (see _runs_beside).  Candidates that share
    the current stream's hardware queue are kept alive while the next one is tried (so the runtime hands out another
    slot) and dropped afterwards.  Falls back to the first candidate -- correct, only without the overlap -- when none
    passes.  Returns (stream, number of candidates rejected or -1)."""
    main = torch.cuda.current_stream(engine.device)
    rejected = []
    for _ in range(tries):
        s = torch.cuda.Stream(device=engine.device)
        if _runs_beside(engine, main, s):
            return s, len(rejected)
        rejected.append(s)
    return rejected[0], -1


class Plan:
    """Launch-plan workspace of the two-launch linearisation (include/mbrl.h: mbrl_fd_plan): a grow-only device buffer
    and the batch size its current plan was made for."""

    def __init__(self):
        self.buf, self.units, self.capacity = None, -1, 0

    def reserve(self, engine, N):
        need = engine.plan_bytes(N)
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(need, dtype=torch.uint8, device=engine.device)   # torch blocks are 512-byte aligned
            self.units = -1
        self.capacity = max(self.capacity, int(N))
        return self.buf


class FdEngine:
    """One immutable model handle on one GPU (reference: build_agent, mujoco/build.py:6-37)."""

    def __init__(self, env, frame_skip=0, device=None):
        if env not in modelspec.MODELS:
            raise KeyError("unknown MUJOCO.ENV %r" % (env,))
        if not torch.cuda.is_available():
            raise lib.MbrlError("no GPU visible: the FD path runs only on the HIP device (no CPU fallback)")
        self.L = lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = C.c_void_p()
        lib.check(self.L.mbrl_model_create(env.encode(), int(frame_skip), self.device.index or 0, C.byref(h)))
        self.h = h
        nq, nv, nu, dt = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        lib.check(self.L.mbrl_model_dims(h, nq, nv, nu, dt))
        self.env, self.nq, self.nv, self.nu, self.dt = env, nq.value, nv.value, nu.value, dt.value
        self.n = self.nq + self.nv
        self._plans = {}                       # HIP stream -> Plan: one grow-only launch-plan buffer per stream that
                                               # linearises through this engine (streams never share a plan)
        self._schedule = {}                    # (mode, nsteps, closed loop, size class) -> planned schedule is faster

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.L.mbrl_model_destroy(h)

    # ---- helpers ----
    def _in(self, t, width):
        t = torch.as_tensor(t, dtype=torch.float64, device=self.device)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        if t.shape[-1] != width:
            raise ValueError("expected last dim %d, got %s" % (width, tuple(t.shape)))
        return t.reshape(-1, width).contiguous()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def side_stream(self):
        """A second HIP stream of this engine's device (rollout.Linearisation overlaps the FD launches with the serial
        step launches on it, bench.py the next step's plan launch with the perturbed rollouts).  Default priority: a
        high-priority stream gained the bench 0.3 %, and with the linearisation on one a training run stopped making
        progress -- not worth it."""
        if getattr(self, "_side", None) is None:
            # not just any new stream: one that is seen to run beside the caller's.  (Found with MBRL_FORCE_DIST=1: once
            # an RCCL communicator exists, the next stream torch hands out shares the default stream's hardware queue,
            # the plan launch no longer overlapped the perturbed rollouts and the bench lost 3 % -- on every rank of a
            # multi-GPU run.)  side_stream_rejected: candidates that were seen to serialise, -1: none overlapped.
            with torch.cuda.device(self.device):
                self._side, self.side_stream_rejected = concurrent_stream(self)
        return self._side

    def _new(self, *shape, dtype=torch.float64):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # ---- entry points ----
    def step(self, qpos, qvel, ctrl):
        qpos, qvel, ctrl = self._in(qpos, self.nq), self._in(qvel, self.nv), self._in(ctrl, self.nu)
        N = qpos.shape[0]
        if qvel.shape[0] != N or ctrl.shape[0] != N:
            raise ValueError("ragged batch")
        qn, vn, r = self._new(N, self.nq), self._new(N, self.nv), self._new(N)
        st = self._new(N, dtype=torch.int32)
        with torch.cuda.device(self.device):
            lib.check(self.L.mbrl_step(self.h, N, _ptr(qpos), _ptr(qvel), _ptr(ctrl), _ptr(qn), _ptr(vn), _ptr(r),
                                       _ptr(st), self._stream()))
        return qn, vn, r, st

    def step_into(self, qpos, qvel, ctrl, qpos_next, qvel_next, reward, status):
        """mbrl_step on caller-owned contiguous float64 / int32 device tensors: no allocation, no copy, no checks
        beyond shape and layout (the rollout loop calls this once per env step)."""
        N = qpos.shape[0]
        for t, w in ((qpos, self.nq), (qvel, self.nv), (ctrl, self.nu), (qpos_next, self.nq), (qvel_next, self.nv)):
            if t.shape != (N, w) or not t.is_contiguous() or t.dtype != torch.float64:
                raise ValueError("step_into: expected a contiguous float64 [%d, %d] tensor, got %s %s" %
                                 (N, w, tuple(t.shape), t.dtype))
        if reward.shape != (N,) or status.shape != (N,) or not reward.is_contiguous() or not status.is_contiguous():
            raise ValueError("step_into: reward / status must be contiguous [N]")
        with torch.cuda.device(self.device):
            lib.check(self.L.mbrl_step(self.h, N, _ptr(qpos), _ptr(qvel), _ptr(ctrl), _ptr(qpos_next), _ptr(qvel_next),
                                       _ptr(reward), _ptr(status), self._stream()))

    def step_record(self, qpos, qvel, ctrl, qpos_next, qvel_next, reward, status, plan, plan_units, unit_offset):
        """step_into that also records these units' rollouts (keys, next states) as units [unit_offset, unit_offset + N)
        of `plan`, a launch plan for plan_units units (mbrl_step_record): the batched rollout's own step launches are
        the nominal rollouts of the linearisation that follows."""
        N = qpos.shape[0]
        buf = plan.buf
        if buf is None or plan.capacity < plan_units:
            raise lib.MbrlError("step_record: reserve the plan for %d units first" % plan_units)
        with torch.cuda.device(self.device):
            lib.check(self.L.mbrl_step_record(self.h, N, _ptr(qpos), _ptr(qvel), _ptr(ctrl), _ptr(qpos_next), _ptr(qvel_next),
                                              _ptr(reward), _ptr(status), _ptr(buf), buf.numel(), int(plan_units),
                                              int(unit_offset), self._stream()))

    def rollout_record(self, Q, V, A, R, S, t0, t1, plan=None, plan_units=0, unit_offset=0):
        """Env steps t0 .. t1-1 of a time-major rollout in ONE launch (mbrl_rollout_record): Q [H+1, B, nq], V [H+1, B, nv],
        A [H, B, nu], R [H, B], S [H, B] are the rollout's own contiguous buffers; step t reads slice t and writes slice
        t + 1.  With `plan` the steps record their units [unit_offset + (t - t0) B, ...) into it like step_record."""
        B, T = Q.shape[1], int(t1 - t0)
        for t, w in ((Q, self.nq), (V, self.nv), (A, self.nu)):
            if t.dim() != 3 or t.shape[1] != B or t.shape[2] != w or not t.is_contiguous() or t.dtype != torch.float64:
                raise ValueError("rollout_record: expected contiguous float64 [*, %d, %d] buffers" % (B, w))
        if not (0 <= t0 < t1 <= A.shape[0] and Q.shape[0] > t1 and V.shape[0] > t1 and R.shape[0] >= t1 and S.shape[0] >= t1):
            raise ValueError("rollout_record: steps [%d, %d) outside the buffers" % (t0, t1))
        if not (R.is_contiguous() and S.is_contiguous() and R.shape[1] == B and S.shape[1] == B):
            raise ValueError("rollout_record: reward / status must be contiguous [H, B]")
        buf = plan.buf if plan is not None else None
        if plan is not None and (buf is None or plan.capacity < plan_units):
            raise lib.MbrlError("rollout_record: reserve the plan for %d units first" % plan_units)
        with torch.cuda.device(self.device):
            lib.check(self.L.mbrl_rollout_record(self.h, B, T, _ptr(Q[t0]), _ptr(V[t0]), _ptr(A[t0]), _ptr(R[t0]), _ptr(S[t0]),
                                                 _ptr(buf), buf.numel() if buf is not None else 0, int(plan_units),
                                                 int(unit_offset), self._stream()))

    def plan_finish(self, plan, plan_units, reward):
        """Header + unit order of a plan recorded by step_record launches (mbrl_fd_plan_finish); `reward` [plan_units]
        is what those launches wrote.  fd_planned(..., plan=plan) runs on it."""
        with torch.cuda.device(self.device):
            lib.check(self.L.mbrl_fd_plan_finish(self.h, int(plan_units), _ptr(reward), _ptr(plan.buf), plan.buf.numel(),
                                                 self._stream()))
        plan.units = int(plan_units)

    def perturbed_policy_actions(self, qpos, qvel, policy, eps=1e-6, fd_mode=lib.FD_FORWARD):
        """Closed-loop mode (cfg.MODEL.POLICY.NETWORK, backward.py:418-419,460-461): the policy's action at every
        state-perturbed copy of every unit, in backward.py's order (qvel entries, then qpos entries).  The state
        reaches the policy as float32 and the action comes back as float64, exactly as the reference casts them.
        Returns ctrl_pert [N, nv+nq, nu] (forward) or [N, nv+nq, 2, nu] (central)."""
        N, n, nq = qpos.shape[0], self.n, self.nq
        s = torch.cat([qpos, qvel], dim=1)
        col = torch.tensor([nq + j for j in range(self.nv)] + list(range(nq)), device=self.device)
        row = torch.arange(n, device=self.device)
        # float32(s + eps e_j) differs from float32(s) in entry j only: copy the float32 state n times and overwrite
        # the n perturbed entries with their float64 sum rounded once (a quarter of the memory traffic of forming the
        # [N, n, n] float64 tensor and casting it; same bits)
        s32 = s.float()
        if fd_mode == lib.FD_CENTRAL:
            P = s32[:, None, None, :].expand(N, n, 2, n).contiguous()           # [N, n, 2, n]
            P[:, row, 0, col] = (s[:, col] + eps).float()
            P[:, row, 1, col] = (s[:, col] - eps).float()
        else:
            P = s32[:, None, :].expand(N, n, n).contiguous()                    # [N, n, n]
            P[:, row, col] = (s[:, col] + eps).float()
        # a strategy object offers its deterministic mean (exploration noise re-drawn per perturbed copy would
        # swamp an eps = 1e-6 difference); a plain callable / nn.Module is used as it is
        fn = getattr(policy, "mean_action", policy)
        with torch.no_grad():
            a = fn(P.reshape(-1, n)).double()
        return a.reshape(P.shape[:-1] + (self.nu,)).contiguous()

    def plan_bytes(self, N):
        need = int(self.L.mbrl_fd_plan_bytes(self.h, N))
        if need < 0:
            lib.check(need)
        return need

    def new_plan(self, N=0):
        """A caller-owned launch plan (workspace of mbrl_fd_plan / mbrl_fd_jacobians_planned) for batches of up to N
        units; it grows on demand.  Pass it as `plan=` to fd_plan / fd_planned / fd_jacobians.  Callers that linearise
        on several streams at once use one Plan per stream (the default, plan=None, does exactly that)."""
        p = Plan()
        if N:
            p.reserve(self, N)
        return p

    def _stream_plan(self):
        """The plan that belongs to the CURRENT stream.  Its buffer is allocated while that stream is current, so the
        caching allocator ties the block to it: a regrown buffer's old block is not handed out again before the
        launches already queued on the stream have run."""
        key = torch.cuda.current_stream(self.device).cuda_stream
        p = self._plans.get(key)
        if p is None:
            p = self._plans[key] = Plan()
        return p

    @property
    def _plan(self):
        """Buffer of the current stream's plan (tests look inside it)."""
        return self._stream_plan().buf

    def _fd_args(self, qpos, qvel, ctrl, eps, fd_mode, policy, ctrl_pert):
        qpos, qvel, ctrl = self._in(qpos, self.nq), self._in(qvel, self.nv), self._in(ctrl, self.nu)
        N, n, nu = qpos.shape[0], self.n, self.nu
        if qvel.shape[0] != N or ctrl.shape[0] != N:
            raise ValueError("ragged batch")
        if policy is not None and ctrl_pert is None and N:
            ctrl_pert = self.perturbed_policy_actions(qpos, qvel, policy, eps, fd_mode)
        if ctrl_pert is not None:
            want = N * n * nu * (2 if fd_mode == lib.FD_CENTRAL else 1)
            ctrl_pert = torch.as_tensor(ctrl_pert, dtype=torch.float64, device=self.device).contiguous()
            if ctrl_pert.numel() != want:
                raise ValueError("ctrl_pert has %d elements, expected %d" % (ctrl_pert.numel(), want))
        return qpos, qvel, ctrl, ctrl_pert

    def new_outputs(self, N):
        n, nu = self.n, self.nu
        return dict(dsds=self._new(N, n, n), dsda=self._new(N, n, nu), drds=self._new(N, n), drda=self._new(N, nu),
                    s_next=self._new(N, n), reward=self._new(N), status=self._new(N, dtype=torch.int32))

    def fd_plan(self, qpos, qvel, ctrl, nsteps=1, out=None, plan=None):
        """First launch of the two-launch linearisation (mbrl_fd_plan): the nominal rollouts (-> out s_next, reward,
        status) and the order in which mbrl_fd_jacobians_planned walks the units.  The plan lives in `plan` (default:
        the current stream's own plan) until the next fd_plan call on it."""
        qpos, qvel, ctrl, _ = self._fd_args(qpos, qvel, ctrl, 0.0, lib.FD_FORWARD, None, None)
        N = qpos.shape[0]
        o = out or self.new_outputs(N)
        if N:
            plan = plan or self._stream_plan()
            buf = plan.reserve(self, N)
            plan.units = N
            with torch.cuda.device(self.device):
                lib.check(self.L.mbrl_fd_plan(self.h, N, _ptr(qpos), _ptr(qvel), _ptr(ctrl), int(nsteps), _ptr(buf),
                                              buf.numel(), _ptr(o.get("s_next")), _ptr(o.get("reward")),
                                              _ptr(o.get("status")), self._stream()))
        return o

    def fd_planned(self, qpos, qvel, ctrl, out, eps=1e-6, fd_mode=lib.FD_FORWARD, nsteps=1, policy=None, ctrl_pert=None,
                   plan=None):
        """Second launch (mbrl_fd_jacobians_planned): the perturbed rollouts of the units fd_plan saw, in its order."""
        qpos, qvel, ctrl, ctrl_pert = self._fd_args(qpos, qvel, ctrl, eps, fd_mode, policy, ctrl_pert)
        N = qpos.shape[0]
        if N:
            plan = plan or self._stream_plan()
            if plan.buf is None or plan.units != N:
                raise lib.MbrlError("fd_planned: no plan for a batch of %d units (call fd_plan on the same inputs first)" % N)
            with torch.cuda.device(self.device):
                lib.check(self.L.mbrl_fd_jacobians_planned(
                    self.h, N, _ptr(qpos), _ptr(qvel), _ptr(ctrl), _ptr(ctrl_pert), float(eps), int(fd_mode), int(nsteps),
                    _ptr(plan.buf), plan.buf.numel(), _ptr(out["dsds"]), _ptr(out["dsda"]), _ptr(out["drds"]),
                    _ptr(out["drda"]), _ptr(out["status"]), self._stream()))
        return out

    def schedule_key(self, N, fd_mode, nsteps, closed_loop):
        return (int(fd_mode), int(nsteps), bool(closed_loop), max(int(N), 1).bit_length())

    def known_schedule(self, N, fd_mode=lib.FD_FORWARD, nsteps=1, closed_loop=False):
        """True / False once tune_schedule has measured this size class, None before."""
        return self._schedule.get(self.schedule_key(N, fd_mode, nsteps, closed_loop))

    def _faster_schedule(self, qpos, qvel, ctrl, ctrl_pert, eps, fd_mode, nsteps, o, recorded=False):
        """Which of the two schedules is faster for this model, mode and batch size?  They produce the same bits, so
        the choice is made by running both once on the caller's inputs (first call per size class only) and kept.
        The plan pays off when the linearisation is long against one latency-bound launch of nominal rollouts:
        HalfCheetah from ~3e4 units (forward) -- not for the pendulums, nor for small batches.  recorded: the caller's
        step launches record the plan (step_record), so the two-launch schedule costs only the sort + the perturbed
        rollouts."""
        N = qpos.shape[0]
        key = self.schedule_key(N, fd_mode, nsteps, ctrl_pert is not None)
        if key not in self._schedule:
            ms = {}
            plan = self._stream_plan()
            for planned in (False, True):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                if planned and recorded:
                    self.fd_plan(qpos, qvel, ctrl, nsteps, o, plan=plan)
                for rep in range(2):            # the first run warms up (plan buffer, code)
                    if rep:
                        ev[0].record(torch.cuda.current_stream(self.device))
                    if planned and recorded:
                        self.plan_finish(plan, N, o["reward"])
                        self.fd_planned(qpos, qvel, ctrl, o, eps, fd_mode, nsteps, ctrl_pert=ctrl_pert, plan=plan)
                    else:
                        self.fd_jacobians(qpos, qvel, ctrl, eps, fd_mode, nsteps, o, ctrl_pert=ctrl_pert, planned=planned)
                ev[1].record(torch.cuda.current_stream(self.device))
                ev[1].synchronize()
                ms[planned] = ev[0].elapsed_time(ev[1])
            self._schedule[key] = ms[True] < ms[False]
        return self._schedule[key]

    def tune_schedule(self, qpos, qvel, ctrl, eps=1e-6, fd_mode=lib.FD_FORWARD, nsteps=1, policy=None, ctrl_pert=None,
                      recorded=False):
        """Measure both schedules once for this size class on the CURRENT stream (host-blocking: call it at warm-up,
        never from a stream that is meant to run beside another one) and remember the faster."""
        qpos, qvel, ctrl, ctrl_pert = self._fd_args(qpos, qvel, ctrl, eps, fd_mode, policy, ctrl_pert)
        N = qpos.shape[0]
        if N < 1024:
            return False
        return self._faster_schedule(qpos, qvel, ctrl, ctrl_pert, eps, fd_mode, nsteps, self.new_outputs(N), recorded)

    def fd_jacobians(self, qpos, qvel, ctrl, eps=1e-6, fd_mode=lib.FD_FORWARD, nsteps=1, out=None, policy=None,
                     ctrl_pert=None, planned=None, plan=None):
        """calculate_gradients for N units (backward.py:343-480).  planned=True: mbrl_fd_plan then
        mbrl_fd_jacobians_planned; planned=False: the single launch mbrl_fd_jacobians_closed_loop; None (default): the
        faster of the two for this model / mode / batch size (measured once per size class, see tune_schedule;
        "cached": use the remembered choice, single launch when there is none -- never measures).  Same bits."""
        qpos, qvel, ctrl, ctrl_pert = self._fd_args(qpos, qvel, ctrl, eps, fd_mode, policy, ctrl_pert)
        N = qpos.shape[0]
        o = out or self.new_outputs(N)
        if planned == "cached":
            planned = bool(self.known_schedule(N, fd_mode, nsteps, ctrl_pert is not None))
        elif planned is None:
            planned = N >= 1024 and self._faster_schedule(qpos, qvel, ctrl, ctrl_pert, eps, fd_mode, nsteps, o)
        if planned:
            self.fd_plan(qpos, qvel, ctrl, nsteps, o, plan=plan)
            return self.fd_planned(qpos, qvel, ctrl, o, eps, fd_mode, nsteps, ctrl_pert=ctrl_pert, plan=plan)
        with torch.cuda.device(self.device):
            lib.check(self.L.mbrl_fd_jacobians_closed_loop(
                self.h, N, _ptr(qpos), _ptr(qvel), _ptr(ctrl), _ptr(ctrl_pert), float(eps), int(fd_mode), int(nsteps),
                _ptr(o["dsds"]), _ptr(o["dsda"]), _ptr(o["drds"]), _ptr(o["drda"]), _ptr(o["s_next"]),
                _ptr(o["reward"]), _ptr(o["status"]), self._stream()))
        return o

    def vjp_scan(self, jac, c, grad_weights="average", time_major=False, shard=None):
        """Backward recurrence over the stored Jacobians.  Episode-major (default): c = dL/dreward [B, H], unit (b, t) at
        b*H + t, returns da [B, H, nu].  time_major: c [H, B], unit at t*B + b, returns da [H, B, nu].  ds0 [B, n].
        shard = (episode_offset, total_episodes) when these B episodes are one rank's block of a larger batch."""
        gw = {"none": 0, "average": 1, "prioritise": 2}.get(grad_weights, 0)
        c = torch.as_tensor(c, dtype=torch.float64, device=self.device).contiguous()
        B, H = (c.shape[1], c.shape[0]) if time_major else c.shape
        da = self._new(H, B, self.nu) if time_major else self._new(B, H, self.nu)
        ds0 = self._new(B, self.n)
        with torch.cuda.device(self.device):
            off, tot = shard if shard is not None else (0, B)
            lib.check(self.L.mbrl_vjp_scan_sharded(self.h, B, H, gw, lib.TIME_MAJOR if time_major else lib.EPISODE_MAJOR,
                                                   int(off), int(tot), _ptr(jac["dsds"]), _ptr(jac["dsda"]),
                                                   _ptr(jac["drds"]), _ptr(jac["drda"]), _ptr(c), _ptr(da), _ptr(ds0),
                                                   self._stream()))
        return da, ds0
