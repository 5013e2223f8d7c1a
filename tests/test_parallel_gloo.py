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
