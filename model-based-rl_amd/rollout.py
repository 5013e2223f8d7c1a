"""Batched replacement of the reference's episode/step loops + loss.backward()
(model/engine/trainer.py:53-77).

The reference walks B episodes x H steps one env.step at a time and lets autograd call the FD
procedure B*H times.  Here: H launches of the step kernel advance all B episodes together, ONE
launch of the FD kernel linearises all B*H stored (s_t, a_t), and one scan kernel runs the backward
recurrence of the op chain.
"""
import os

import torch

from . import lib


def _time_major_buffers(engine, s0, H):
    """Q [H+1, B, nq], V [H+1, B, nv], R [H, B], S [H, B]: step t reads slice t and writes slice t+1 in place, so the
    loop allocates and copies nothing."""
    s0 = torch.as_tensor(s0, dtype=torch.float64, device=engine.device)
    B, nq = s0.shape[0], engine.nq
    Q = torch.empty((H + 1, B, nq), dtype=torch.float64, device=engine.device)
    V = torch.empty((H + 1, B, engine.nv), dtype=torch.float64, device=engine.device)
    R = torch.empty((H, B), dtype=torch.float64, device=engine.device)
    S = torch.empty((H, B), dtype=torch.int32, device=engine.device)
    Q[0], V[0] = s0[:, :nq], s0[:, nq:]
    return Q, V, R, S


def _episode_major(Q, V, R, S):
    return torch.cat([Q, V], dim=2).permute(1, 0, 2).contiguous(), R.t().contiguous(), S.t().contiguous()


def rollout(engine, s0, actions):
    """s0 [B, n], actions [B, H, nu] -> states [B, H+1, n], rewards [B, H], status [B, H]."""
    B, H, _ = actions.shape
    Q, V, R, S = _time_major_buffers(engine, s0, H)
    A = torch.as_tensor(actions, dtype=torch.float64, device=engine.device).permute(1, 0, 2).contiguous()   # [H, B, nu]
    for t in range(H):
        engine.step_into(Q[t], V[t], A[t], Q[t + 1], V[t + 1], R[t], S[t])
    return _episode_major(Q, V, R, S)


def action_gradients(engine, states, actions, c, grad_weights="average", eps=1e-6, fd_mode=lib.FD_FORWARD):
    """dL/d actions [B, H, nu] and dL/d s0 [B, n] for L = sum_{b,t} c[b,t] * r[b,t], with the op
    weighting of model/blocks/mujoco.py:46-76.  One FD launch over B*H units + one scan."""
    B, H, nu = actions.shape
    nq = engine.nq
    s = states[:, :H].reshape(B * H, engine.n)
    jac = engine.fd_jacobians(s[:, :nq], s[:, nq:], actions.reshape(B * H, nu), eps=eps, fd_mode=fd_mode)
    da, ds0 = engine.vjp_scan(jac, c, grad_weights)
    return da, ds0, jac


class Linearisation:
    """The Jacobians of every (s_t, a_t) of a batched rollout, TIME-MAJOR (unit (b, t) at t*B + b), produced on a second
    HIP stream while the rollout is still running.

    A step launch of B episodes occupies B/64 waves -- 6 % of the chip for B = 4096 -- and the H of them are serial.
    The linearisation of time steps [t0, t1) needs only their states and actions, which exist as soon as step t1 - 2
    has finished, so it is submitted chunk by chunk to a side stream and fills the other SIMDs; by the time the last
    step is done only the last chunk is outstanding.  The reference has no counterpart (its FD runs inside
    loss.backward(), trainer.py:77); the results are the same Jacobians."""

    def __init__(self, engine, B, H, eps, fd_mode, policy, chunk, nsteps=1):
        self.eng, self.B, self.H, self.eps, self.fd_mode, self.policy = engine, B, H, eps, fd_mode, policy
        self.nsteps = int(nsteps)   # cfg.MODEL.NSTEPS_FOR_BACKWARD (backward.py:359): env steps per FD rollout
        self.chunk = max(1, int(chunk))
        self.jac = engine.new_outputs(B * H)
        self.stream = engine.side_stream()
        self.next_t = 0
        self._recorded = False
        self._chunk_plans = {}      # chunk start t0 -> Plan being recorded by the rollout's own step launches
        # the Jacobian buffers were allocated on the caller's stream and are written on the side stream: tell the
        # caching allocator, so that their blocks are not handed out again while a side-stream launch may still run
        for t in self.jac.values():
            t.record_stream(self.stream)

    def _chunk(self, t):
        t0 = (t // self.chunk) * self.chunk
        return t0, min(self.H, t0 + self.chunk)

    def recorder(self, t):
        """(plan, plan_units, unit_offset) when the step launch of time t should record its rollouts into its chunk's
        launch plan (FdEngine.step_record), else None.  The step launches ARE the nominal rollouts of the linearisation
        (the reference's forward op hands calculate_gradients its next_state, model/blocks/mujoco.py:20-31), so a chunk
        that will run the two-launch schedule needs no nominal launch of its own -- only the sort."""
        t0, t1 = self._chunk(t)
        units = (t1 - t0) * self.B
        if os.environ.get("MBRL_NO_STEP_RECORD"):      # measurement knob (tools/train_throughput.py)
            return None
        if self.nsteps != 1:         # a step launch is the nominal rollout of a ONE-env-step linearisation only
            return None
        if not self.eng.known_schedule(units, self.fd_mode, 1, self.policy is not None):
            return None
        plan = self._chunk_plans.get(t0)
        if plan is None:
            cache = self.eng.__dict__.setdefault("_rollout_plans", {})
            plan = cache.setdefault((t0, self.chunk), self.eng.new_plan())
            plan.reserve(self.eng, units)
            plan.buf.record_stream(self.stream)
            plan.units = -1
            self._chunk_plans[t0] = plan
        return plan, units, (t - t0) * self.B

    def submit(self, Q, V, A, t_done, R=None, S=None):
        """Called on the rollout's stream after step t_done - 1 was enqueued: linearise the finished chunks.  R, S: the
        rollout's reward / status buffers (needed for chunks whose plan the step launches recorded)."""
        if not self._recorded:      # the rollout's state / action buffers are read on the side stream
            for t in (Q, V, A):
                t.record_stream(self.stream)
            self._recorded = True
        while self.next_t < self.H and (self.next_t + self.chunk <= t_done or t_done >= self.H):
            t0, t1 = self.next_t, min(self.H, self.next_t + self.chunk)
            self.next_t = t1
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.eng.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                sl = slice(t0 * self.B, t1 * self.B)
                out = {k: v[sl] for k, v in self.jac.items()}
                e = self.eng
                plan = self._chunk_plans.pop(t0, None)
                if plan is not None:
                    # the step launches of this chunk recorded keys and next states: finish the plan (sort), then the
                    # perturbed rollouts only
                    e.plan_finish(plan, (t1 - t0) * self.B, R[t0:t1].reshape(-1))
                    out["status"].copy_(S[t0:t1].reshape(-1))
                    out["reward"].copy_(R[t0:t1].reshape(-1))
                    out["s_next"].copy_(torch.cat([Q[t0 + 1:t1 + 1], V[t0 + 1:t1 + 1]], dim=2).reshape(-1, e.n))
                    e.fd_planned(Q[t0:t1].reshape(-1, e.nq), V[t0:t1].reshape(-1, e.nv), A[t0:t1].reshape(-1, e.nu), out,
                                 eps=self.eps, fd_mode=self.fd_mode, policy=self.policy, plan=plan)
                    continue
                # schedule: the remembered choice for this size class (single launch until tune() has measured it) --
                # the measurement blocks the host and must not run here, beside the rollout.  The plan, if any, is the
                # side stream's own (FdEngine keeps one per stream).
                e.fd_jacobians(Q[t0:t1].reshape(-1, e.nq), V[t0:t1].reshape(-1, e.nv), A[t0:t1].reshape(-1, e.nu),
                               eps=self.eps, fd_mode=self.fd_mode, nsteps=self.nsteps, out=out, policy=self.policy,
                               planned="cached")

    def tune(self, Q, V, A):
        """Once per size class, on the caller's stream with the side stream drained (RolloutRewards.backward calls it
        after wait()): measure which schedule is faster for a chunk of this rollout, for the following epochs."""
        e, n = self.eng, min(self.chunk, self.H) * self.B
        if n >= 1024 and e.known_schedule(n, self.fd_mode, self.nsteps, self.policy is not None) is None:
            k = min(self.chunk, self.H)
            e.tune_schedule(Q[:k].reshape(-1, e.nq), V[:k].reshape(-1, e.nv), A[:k].reshape(-1, e.nu), eps=self.eps,
                            fd_mode=self.fd_mode, nsteps=self.nsteps, policy=self.policy, recorded=self.nsteps == 1)

    def __del__(self):
        # the side stream reads buffers this object keeps alive (lin.keep): never let them return to the allocator
        # while a launch that was submitted but not waited for (no backward pass) may still be running
        try:
            self.stream.synchronize()
        except Exception:
            pass

    def wait(self):
        """Make the current stream wait for everything submitted.  MBRL_FD_WATCHDOG_S=<seconds> (off by default: it
        blocks the host) turns a side stream that stops making progress into an error that names what was outstanding,
        instead of a run that sits in its next host synchronisation for ever ("the stalled run")."""
        done = torch.cuda.Event()
        done.record(self.stream)
        limit = float(os.environ.get("MBRL_FD_WATCHDOG_S", "0") or 0)
        if limit > 0:
            import time
            t0 = time.monotonic()
            while not done.query():
                if time.monotonic() - t0 > limit:
                    raise lib.MbrlError(
                        "linearisation side stream made no progress for %.0f s: %d of %d time steps submitted in chunks of "
                        "%d, main stream %#x, side stream %#x (rejected candidates: %s)" %
                        (limit, self.next_t, self.H, self.chunk, torch.cuda.current_stream(self.eng.device).cuda_stream,
                         self.stream.cuda_stream, getattr(self.eng, "side_stream_rejected", None)))
                time.sleep(0.001)
        torch.cuda.current_stream(self.eng.device).wait_event(done)


class RolloutRewards(torch.autograd.Function):
    """rewards [B, H] of a stored rollout as a differentiable function of (actions [B, H, nu], s0 [B, n]).

    Forward hands back the rewards the step kernel produced; backward is the whole chain of MjBlock.backward
    calls autograd would issue for the epoch (model/blocks/mujoco.py:43-104) in two launches: ONE FD launch over
    the B*H stored (s_t, a_t) and one scan.  Closed-loop mode passes `policy` (cfg.MODEL.POLICY.NETWORK,
    backward.py:418-419,460-461).  `grad_weights` is cfg.MODEL.POLICY.GRAD_WEIGHTS."""

    @staticmethod
    def forward(ctx, actions, s0, engine, states, rewards, grad_weights, eps, fd_mode, policy, raise_on_status, lin=None,
                shard=None, nsteps=1):
        ctx.nsteps = int(nsteps)  # cfg.MODEL.NSTEPS_FOR_BACKWARD
        ctx.engine, ctx.gw, ctx.eps, ctx.fd_mode, ctx.policy, ctx.chk = engine, grad_weights, eps, fd_mode, policy, raise_on_status
        ctx.shard = shard        # (first episode, episodes of the whole batch) of a sharded run, or None
        ctx.lin = lin            # Linearisation already under way on the side stream (rollout_with_policy), or None
        ctx.save_for_backward(actions.detach(), states)
        # the reference's reward op rounds through fp32 (torch.Tensor([r]).double(), mujoco.py:38); kept
        return rewards.float().double()

    @staticmethod
    def backward(ctx, c):
        actions, states = ctx.saved_tensors
        eng = ctx.engine
        B, H, nu = actions.shape
        if ctx.lin is not None:
            ctx.lin.wait()
            jac = ctx.lin.jac
            keep = getattr(ctx.lin, "keep", None)
            if keep is not None:
                ctx.lin.tune(*keep)
        else:
            s = states[:, :H].reshape(B * H, eng.n)
            jac = eng.fd_jacobians(s[:, :eng.nq], s[:, eng.nq:], actions.reshape(B * H, nu), eps=ctx.eps,
                                   fd_mode=ctx.fd_mode, nsteps=ctx.nsteps, policy=ctx.policy)
        if ctx.chk:
            bad = int((jac["status"] != 0).sum())
            if bad:
                raise lib.MbrlError("FD linearisation flagged %d of %d units (status bits, include/mbrl.h)" % (bad, B * H))
        else:
            # Flagged units (MBRL_ST_*): their Jacobians are not to be trusted.  Cut the episode's gradient chain there:
            # zero Jacobians and zero loss weight for the flagged unit, so nothing flows through or out of it (the
            # trainer masks the same episodes' losses, as the reference's NaN-loss rule does, strategies.py:191-192).
            flagged = jac["status"] != 0
            if bool(flagged.any()):
                jac = dict(jac)
                for k in ("dsds", "dsda", "drds", "drda"):
                    jac[k] = jac[k].masked_fill(flagged.reshape((-1,) + (1,) * (jac[k].dim() - 1)), 0.0)
        if ctx.lin is not None:
            da, ds0 = eng.vjp_scan(jac, c.t().contiguous(), ctx.gw, time_major=True, shard=ctx.shard)
            da = da.permute(1, 0, 2)
        else:
            da, ds0 = eng.vjp_scan(jac, c.contiguous(), ctx.gw, shard=ctx.shard)
        return da, ds0, None, None, None, None, None, None, None, None, None, None, None


def rollout_with_policy(engine, agent, model_policy, s0, horizon, linearise=None):
    """Forward pass of one epoch for all episodes together (trainer.py:57-74 + basic.py:29-46): the policy sees
    the DETACHED state (basic.py:38), the step kernel advances every episode.  Returns states [B, H+1, n],
    actions [B, H, nu] (with their autograd graph to the policy parameters), rewards [B, H], status [B, H] and --
    with linearise = dict(eps, fd_mode, policy, chunk) -- the Linearisation running beside the rollout (else None)."""
    Q, V, R, S = _time_major_buffers(engine, s0, horizon)
    B = Q.shape[1]
    A = torch.empty((horizon, B, engine.nu), dtype=torch.float64, device=engine.device)
    lin = Linearisation(engine, B, horizon, **linearise) if linearise else None
    # an open-loop policy's actions do not depend on the states: all of them at once, so that nothing but the step
    # launches sits between two env steps (strategies.forward_open_loop; None: ask the policy step by step)
    ahead = getattr(model_policy, "forward_open_loop", None)
    ahead = ahead(horizon, B) if ahead is not None and os.environ.get("MBRL_NO_ACTIONS_AHEAD") is None else None
    if ahead is not None:
        A.copy_(ahead.detach())
    acts = []
    # actions known ahead: the env steps of a whole chunk run in ONE launch (FdEngine.rollout_record).  Each wave keeps
    # its episodes across the steps, so nothing waits for every step's slowest wave; same bits as step by step.
    multi = ahead is not None and os.environ.get("MBRL_NO_MULTISTEP") is None and hasattr(engine, "rollout_record")
    if multi:
        span = lin.chunk if lin is not None else horizon
        for t0 in range(0, horizon, span):
            t1 = min(horizon, t0 + span)
            rec = lin.recorder(t0) if lin is not None else None
            if rec is not None:
                engine.rollout_record(Q, V, A, R, S, t0, t1, plan=rec[0], plan_units=rec[1], unit_offset=rec[2])
                for t in range(t0 + 1, t1):          # the chunk's plan is the same object for every step of it
                    lin.recorder(t)
            else:
                engine.rollout_record(Q, V, A, R, S, t0, t1)
            if lin is not None:
                for buf in ((R, S) if t0 == 0 else ()):
                    buf.record_stream(lin.stream)
                lin.submit(Q, V, A, t1, R, S)
    for t in (range(0) if multi else range(horizon)):
        agent.set_step_idx(t)
        if ahead is None:
            a = model_policy(torch.cat([Q[t], V[t]], dim=1).float()).double()
            acts.append(a)
            A[t].copy_(a.detach())
        rec = lin.recorder(t) if lin is not None else None
        if rec is not None:
            engine.step_record(Q[t], V[t], A[t], Q[t + 1], V[t + 1], R[t], S[t], *rec)
        else:
            engine.step_into(Q[t], V[t], A[t], Q[t + 1], V[t + 1], R[t], S[t])
        if lin is not None:
            for buf in ((R, S) if t == 0 else ()):
                buf.record_stream(lin.stream)
            lin.submit(Q, V, A, t + 1, R, S)
    states, rewards, status = _episode_major(Q, V, R, S)
    agent.set_step_idx(horizon)
    if lin is not None:
        lin.keep = (Q, V, A)     # the side stream reads these: they live as long as the linearisation does
    actions = ahead.permute(1, 0, 2) if ahead is not None else torch.stack(acts, dim=1)
    return states, actions, rewards, status, lin
