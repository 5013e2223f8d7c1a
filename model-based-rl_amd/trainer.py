"""Batched trainer (reference: model/engine/trainer.py:12-146).

Same contract as the reference's `do_training(cfg, logger, output_results_dir, output_rec_dir, output_weights_dir,
iter)`: seeds (:21-23), agent + model (:26-31), EPOCHS x [BATCH_SIZE episodes x MAX_HORIZON_STEPS steps], the loss
tensor `[BATCH_SIZE, MAX_HORIZON_STEPS]` fp64 pre-filled with NaN (:54-55), `agent.running_sum = 0` (:76),
`policy_net.optimize(batch_loss)` (:77), the per-epoch log line (:112), `state_dict` checkpoints every
LOG.CHECKPOINT_PERIOD epochs (:117-119) and the CSV `epoch, objective_loss, average_sd` (:50,87-89,144).
What differs is the schedule: the reference walks episodes and steps one `env.step` at a time and lets autograd
call the FD procedure B*H times; here H step launches advance all episodes, ONE FD launch linearises the B*H
stored (s_t, a_t) and one scan runs the backward recurrence (rollout.RolloutRewards).  cfg.MODEL.ENGINE_PATH =
"ops" runs the reference's op-by-op graph instead (Basic + MjBlock ops, one FD launch per step).  Resume from
cfg.MODEL.WEIGHTS (commented out in the reference, model/build.py:8-9) is implemented.
Multi-GPU: episodes are sharded over ranks (parallel.shard_range), the policy gradient is summed with one
flattened all-reduce per epoch; rank 0 logs and writes files.  Viewer / video / visdom are out of scope.
"""
import csv
import logging
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from . import lib, parallel, rollout
from .agent import build_agent
from .basic import Basic
from .strategies import build_policy


def setup_logger(name, save_dir, txt_file_name="log"):
    """utils/logger.py:7-22."""
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    logger.handlers.clear()
    fmt = logging.Formatter("%(asctime)s %(name)s %(levelname)s: %(message)s")
    ch = logging.StreamHandler(stream=sys.stdout)
    ch.setFormatter(fmt)
    logger.addHandler(ch)
    if save_dir:
        fh = logging.FileHandler(os.path.join(save_dir, "{}.txt".format(txt_file_name)))
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    return logger


def save_dict_into_csv(save_dir, file_name, output):
    """utils/logger.py:25-35."""
    with open(os.path.join(save_dir, file_name), "w") as fh:
        w = csv.writer(fh)
        w.writerow(output.keys())
        w.writerows(zip(*output.values()))


def build_model(cfg, agent, shard=None):
    """model/build.py:5-10 (+ the resume the reference left commented out)."""
    model = Basic(cfg, agent, policy_net=build_policy(cfg, agent, shard=shard))
    if cfg.MODEL.WEIGHTS:
        model.load_state_dict(torch.load(cfg.MODEL.WEIGHTS, map_location=agent.engine.device, weights_only=True))
    return model


def do_training(cfg, logger, output_results_dir, output_rec_dir, output_weights_dir, iter, device=None, engine=None):
    """`engine`: an object with FdEngine's interface to use instead of creating one on `device` (the world-size-2 CPU
    test passes a stand-in; the product path always creates the HIP engine)."""
    grouped = parallel.active()            # a process group exists (possibly of one rank: MBRL_FORCE_DIST)
    rank = dist.get_rank() if grouped else 0
    world = dist.get_world_size() if grouped else 1
    if cfg.MODEL.RANDOM_SEED > 0:                                   # trainer.py:21-23
        np.random.seed(cfg.MODEL.RANDOM_SEED + iter)
        torch.manual_seed(cfg.MODEL.RANDOM_SEED + iter)

    B, H = cfg.MODEL.BATCH_SIZE, cfg.MODEL.POLICY.MAX_HORIZON_STEPS
    lo, hi = parallel.shard_range(B, rank, world)                   # this rank's episodes
    if hi - lo < 1:
        raise ValueError("MODEL.BATCH_SIZE %d is smaller than the number of ranks %d" % (B, world))
    agent = build_agent(cfg, batch=hi - lo, device=device, engine=engine, shard=(lo, B))
    model = build_model(cfg, agent, shard=(lo, hi))                 # same seed on every rank: identical parameters
    model.train()
    policy, eng = model.policy_net, agent.engine
    fd_mode = lib.FD_CENTRAL if str(cfg.MODEL.get("FD_MODE", "forward")) == "central" else lib.FD_FORWARD
    path = str(cfg.MODEL.get("ENGINE_PATH", "fused"))
    output = {"epoch": [], "objective_loss": [], "average_sd": []}

    def grad_sync(grads):
        parallel.allreduce_sum_(grads)

    for epoch_idx in range(cfg.MODEL.EPOCHS):
        batch_loss = torch.full((hi - lo, H), float("nan"), dtype=torch.float64, device=eng.device)   # trainer.py:54-55
        s0 = agent.reset()
        s0 = s0.reshape(agent.batch, -1)
        if path == "ops":                                           # the reference's op-by-op graph, batched
            state = s0
            for t in range(H):
                state, reward = model(state)
                batch_loss[:, t] = -reward.reshape(-1)
        else:
            eps = float(cfg.MODEL.get("FD_EPS", 1e-6))
            fd_policy = policy if cfg.MODEL.POLICY.NETWORK else None
            overlap = int(cfg.MODEL.get("FD_OVERLAP_STEPS", 4))     # 0: linearise inside loss.backward(), as the reference
            states, actions, rewards, status, lin = rollout.rollout_with_policy(
                eng, agent, policy, s0, H,
                linearise=dict(eps=eps, fd_mode=fd_mode, policy=fd_policy, chunk=overlap,
                               nsteps=int(cfg.MODEL.NSTEPS_FOR_BACKWARD)) if overlap > 0 else None)
            r = rollout.RolloutRewards.apply(actions, s0, eng, states, rewards, cfg.MODEL.POLICY.GRAD_WEIGHTS,
                                             eps, fd_mode, fd_policy, False, lin, (lo, B), int(cfg.MODEL.NSTEPS_FOR_BACKWARD))
            batch_loss[:] = -r
            # A step the simulator flagged (MBRL_ST_*: non-finite state, solver at its iteration cap, ...) ends that
            # episode's contribution: its loss from there on is NaN, which `optimize` turns into zero loss and zero
            # gradient (the reference's own rule, strategies.py:191-192); mujoco-py would have raised instead.
            dead = (status != 0).cumsum(dim=1) > 0
            if bool(dead.any()):
                batch_loss = batch_loss.masked_fill(dead, float("nan"))
                if rank == 0:
                    logger.warning("epoch %d: %d of %d episodes masked from their first flagged step (status bits, "
                                   "include/mbrl.h)" % (epoch_idx, int(dead.any(dim=1).sum()), hi - lo))
        agent.running_sum = 0                                       # trainer.py:76
        loss = policy.optimize(batch_loss, grad_sync=grad_sync if grouped else None)   # trainer.py:77
        if grouped:     # logged objective = batch mean of episode losses over ALL ranks' episodes
            t = torch.tensor([loss["objective_loss"] * (hi - lo)], dtype=torch.float64, device=eng.device)
            dist.all_reduce(t)
            loss["objective_loss"] = float(t.item()) / B

        output["objective_loss"].append(loss["objective_loss"])
        output["epoch"].append(epoch_idx)
        output["average_sd"].append(float(np.mean(policy.get_clamped_sd())))
        if rank == 0 and epoch_idx % cfg.LOG.PERIOD == 0:
            logger.info("REWARD: \t\t{} (iteration {})".format(loss["objective_loss"], epoch_idx))
        if rank == 0 and epoch_idx % cfg.LOG.CHECKPOINT_PERIOD == 0:
            torch.save(model.state_dict(), os.path.join(output_weights_dir, "iter_{}.pth".format(epoch_idx)))

    if rank == 0:
        save_dict_into_csv(output_results_dir, "output_{}".format(iter), output)
    return agent
