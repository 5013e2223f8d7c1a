"""CPU, world_size 2 over gloo: the N>1 path -- episodes sharded over ranks with no data-path collective, one
flattened all-reduce on the shared action-sequence gradient -- reproduces the single-process result.  Per-rank
compute here is the CPU oracle + the numpy recurrence (the checker); on the GPU box the same host code drives
the HIP path."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _grads(lo, hi):
    """d(sum_b sum_t r)/d(shared open-loop actions [H, nu]) from episodes [lo, hi) of a fixed synthetic batch."""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import Oracle
    from test_oracle_golden import recurrence
    env, B, H = "InvertedDoublePendulumEnv", 6, 4
    o = Oracle(env)
    rng = np.random.default_rng(3)
    s0 = np.concatenate([rng.uniform(-0.05, 0.05, (B, o.nq)), 0.1 * rng.standard_normal((B, o.nv))], 1)
    acts = rng.uniform(-0.5, 0.5, (H, o.nu))           # shared by every episode
    g = np.zeros((H, o.nu))
    for b in range(lo, hi):
        s, S = s0[b], []
        for t in range(H):
            S.append(s)
            q, v, _, _ = o.step(s[:o.nq], s[o.nq:], acts[t])
            s = np.concatenate([q[0], v[0]])
        S = np.array(S)
        j = o.fd_jacobians(S[:, :o.nq], S[:, o.nq:], acts)
        gd = dict(dsds=j["dsds"][None], dsda=j["dsda"][None], drds=j["drds"][None], drda=j["drda"][None],
                  c=np.ones((1, H)), actions=acts[None])
        da, _ = recurrence(gd, "average")
        g += da[0]
    return g, B


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import mbrl_amd
    from mbrl_amd import parallel
    _, B = _grads(0, 0)
    lo, hi = parallel.shard_range(B, rank, world)
    g, _ = _grads(lo, hi)
    t = torch.from_numpy(g.copy())
    extra = torch.tensor([float(hi - lo)], dtype=torch.float64)
    parallel.allreduce_sum_([t, extra])
    tmax = parallel.max_over_ranks(1.0 + rank)
    if rank == 0:
        np.savez(out, g=t.numpy(), n=extra.numpy(), tmax=tmax)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    sys.path.insert(0, ROOT)
    import mbrl_amd
    from mbrl_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 163840):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    out = str(tmp_path / "r0.npz")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    ref, B = _grads(0, 6)
    assert got["n"][0] == B and got["tmax"] == 2.0
    np.testing.assert_allclose(got["g"], ref, rtol=1e-12, atol=1e-14)


def _strategy_worker(rank, world, port, out):
    """Sharded optimisation step of the in-repo strategy: each rank owns a block of episodes, weights its loss by
    local_batch / BATCH_SIZE, and the gradients meet in ONE flattened all-reduce (trainer.do_training's grad_sync)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import mbrl_amd
    from mbrl_amd import parallel
    from test_strategy_host import strategy_step
    lo, hi = parallel.shard_range(6, rank, world)
    params = strategy_step((lo, hi), grad_sync=parallel.allreduce_sum_)
    if rank == 0:
        np.savez(out, **{k: v.numpy() for k, v in params.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_strategy_step_matches_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_strategy_host import strategy_step
    out = str(tmp_path / "s0.npz")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_strategy_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    ref = strategy_step((0, 6), grad_sync=None)
    for k, v in ref.items():
        np.testing.assert_allclose(got[k], v.numpy(), rtol=1e-12, atol=1e-14)


# ---- trainer.do_training under torch.distributed, end to end (world size 2, gloo) ----
def _train_cfg(weights):
    import mbrl_amd  # noqa: F401
    from mbrl_amd import config
    cfg = config.cfg_for_env("InvertedDoublePendulumEnv")
    cfg.MODEL.BATCH_SIZE, cfg.MODEL.POLICY.MAX_HORIZON_STEPS, cfg.MODEL.EPOCHS = 5, 4, 3     # 5 episodes: shards of 3 + 2
    cfg.MODEL.POLICY.ARCH, cfg.MODEL.POLICY.METHOD = "VariationalOptimization", "R"
    cfg.MODEL.POLICY.INITIAL_SD, cfg.MODEL.RANDOM_SEED = 0.2, 11
    cfg.MODEL.POLICY.GRAD_WEIGHTS = weights
    cfg.MODEL.FD_OVERLAP_STEPS = 0            # the stand-in engine has no streams
    cfg.SOLVER.BASE_LR, cfg.SOLVER.STD_LR_FACTOR, cfg.SOLVER.BIAS_LR_FACTOR = 0.01, 1.0, 1
    cfg.LOG.CHECKPOINT_PERIOD = 2
    return cfg


def _do_training(weights, out_dir):
    """main.train (rank-0 directories, broadcast run name) -> trainer.do_training over the CPU stand-in engine."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import mbrl_amd  # noqa: F401
    from mbrl_amd import main as cli
    from cpu_engine import CpuEngine
    cfg = _train_cfg(weights)
    cfg.OUTPUT.DIR, cfg.OUTPUT.NAME = out_dir, "timestamp"
    agent, run_dir = cli.train(cfg, 1, engine=CpuEngine(cfg.MUJOCO.ENV, cfg.MODEL.FRAME_SKIP))
    return {k: v.detach().clone() for k, v in agent.policy_net.state_dict().items()}, run_dir


def _train_worker(rank, world, port, weights, out_dir, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, run_dir = _do_training(weights, out_dir)
    if rank == 0:
        np.savez(out, run_dir=run_dir, **{k: v.numpy() for k, v in params.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("weights", ["average", "prioritise"])
def test_two_rank_do_training_matches_single_process(weights, tmp_path):
    """The distributed branch of do_training (shard_range, per-shard loss weight, sliced exploration noise, flattened
    gradient all-reduce, logged objective over all ranks, 'prioritise' seeded with the shard's global episode offset)
    trains the parameters a single process trains; only rank 0 writes, and every rank agrees on the run directory."""
    import csv
    out = str(tmp_path / "p.npz")
    port = 33500 + os.getpid() % 2000
    mp.spawn(_train_worker, args=(2, port, weights, str(tmp_path / "two"), out), nprocs=2, join=True)
    got = np.load(out)
    ref, ref_dir = _do_training(weights, str(tmp_path / "one"))
    for k, v in ref.items():
        np.testing.assert_allclose(got[k], v.numpy(), rtol=1e-9, atol=1e-12, err_msg=k)
    run = str(got["run_dir"])
    assert sorted(os.listdir(os.path.join(run, "weights"))) == ["iter_0.pth", "iter_2.pth"]
    rows2 = list(csv.reader(open(os.path.join(run, "results", "output_0"))))
    rows1 = list(csv.reader(open(os.path.join(ref_dir, "results", "output_0"))))
    assert rows2[0] == rows1[0] == ["epoch", "objective_loss", "average_sd"]
    np.testing.assert_allclose([[float(x) for x in r] for r in rows2[1:]], [[float(x) for x in r] for r in rows1[1:]],
                               rtol=1e-9)
