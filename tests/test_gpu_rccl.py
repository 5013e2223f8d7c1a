"""GPU: the RCCL branch of the multi-GPU path, exercised with what a one-GPU box has -- a process group of ONE rank.

`north_star` asks for 1/2/4/8-GPU figures; the driver launches those on an 8-GPU node.  Everything the N > 1 run does
beyond the single-process run -- init_process_group("nccl", device_id=...), barrier(), the ReduceOp.MAX reduce of the
timing, parallel.allreduce_sum_ on device tensors, the broadcast of the run name -- is gated on "a process group exists",
and MBRL_FORCE_DIST=1 creates one with a single rank.  So RCCL is initialised and reduces on the MI355X here, before the
driver's 8-GPU run does it for the first time.  (World-size-2 semantics are covered by the gloo tests,
tests/test_parallel_gloo.py and tests/test_gpu_trainer.py.)  The reference is single-process (model/engine/trainer.py:77)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MBRL_DIST_BACKEND", "MBRL_FORCE_DIST"):
        env.pop(k, None)
    env.update(extra)
    return env


def _bench(env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu",
                          "--no-extra"], env=env, cwd=ROOT, stdout=subprocess.PIPE, timeout=280, check=True).stdout.decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_through_rccl_with_one_rank():
    """bench.py with a forced one-rank `nccl` process group: same line (n_gpus = 1), same rate within 2 %."""
    plain = _bench(_env())
    forced = _bench(_env(MBRL_FORCE_DIST="1", TORCH_DISTRIBUTED_DEBUG="INFO"))
    assert forced["n_gpus"] == 1 and plain["n_gpus"] == 1
    assert forced["units_with_bad_status"] == 0
    assert abs(forced["value"] / plain["value"] - 1.0) < 0.02, (forced["value"], plain["value"])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("weights", ["average", "prioritise"])
def test_trainer_through_rccl_with_one_rank_trains_bit_for_bit(weights, tmp_path):
    """The command line with MBRL_FORCE_DIST=1: init_process_group("nccl"), the broadcast of the run name, the flattened
    gradient all-reduce and the loss reduce run through RCCL on device tensors; summing over one rank changes no bit, so
    the trained parameters equal the plain run's exactly."""
    import torch
    opts = ["--config-file", os.path.join(ROOT, "configs", "half_cheetah.yaml"), "--opts", "MODEL.EPOCHS", "3",
            "MODEL.BATCH_SIZE", "6", "MODEL.POLICY.MAX_HORIZON_STEPS", "5", "MODEL.POLICY.GRAD_WEIGHTS", weights,
            "MODEL.POLICY.INITIAL_SD", "0.2", "MODEL.RANDOM_SEED", "5", "LOG.CHECKPOINT_PERIOD", "2",
            "OUTPUT.DIR", str(tmp_path)]
    subprocess.check_call([sys.executable, "-m", "mbrl_amd.main"] + opts + ["OUTPUT.NAME", "plain"], env=_env(), cwd=ROOT,
                          stdout=subprocess.DEVNULL, timeout=280)
    p = subprocess.run([sys.executable, "-m", "mbrl_amd.main"] + opts + ["OUTPUT.NAME", "rccl"],
                       env=_env(MBRL_FORCE_DIST="1", NCCL_DEBUG="VERSION"), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=280)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    one = torch.load(tmp_path / "HalfCheetahEnv" / "plain" / "weights" / "iter_2.pth", weights_only=True)
    two = torch.load(tmp_path / "HalfCheetahEnv" / "rccl" / "weights" / "iter_2.pth", weights_only=True)
    for k in one:
        assert torch.equal(one[k].cpu(), two[k].cpu()), k
    assert sorted(os.listdir(tmp_path / "HalfCheetahEnv" / "rccl" / "results")) == ["output_0"]


def test_collectives_of_the_path_on_device_tensors():
    """In-process: a one-rank nccl group; parallel.allreduce_sum_ and max_over_ranks run through RCCL on the GPU and
    return their inputs bit for bit."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import mbrl_amd
from mbrl_amd import parallel, launch
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(launch.free_port()), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=launch.init_timeout())
assert dist.get_backend() == "nccl" and parallel.active()
g = [torch.randn(6, 40, dtype=torch.float64, device="cuda"), torch.randn(240, dtype=torch.float64, device="cuda")]
ref = [t.clone() for t in g]
parallel.allreduce_sum_(g)
dist.barrier()
torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(g, ref))
assert parallel.max_over_ranks(1.25) == 1.25
dist.destroy_process_group()
print("rccl ok")
''' % ROOT
    p = subprocess.run([sys.executable, "-c", code], env=_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=200)
    assert p.returncode == 0 and b"rccl ok" in p.stdout, p.stdout.decode()[-2000:]
