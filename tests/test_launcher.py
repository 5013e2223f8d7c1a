"""CPU: the self-launcher of `bench.py --gpus N` / `python -m mbrl_amd.main --gpus N` (mbrl_amd/launch.py) fails fast.
A rank that dies before -- or inside -- a collective must not leave its siblings waiting in it: the launcher stops them
and returns the dead rank's code within seconds (128 + signal for a signal death).  Ranks here are plain CPU processes
over gloo; the launcher is the code the GPU entry points use."""
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mbrl_amd  # noqa: E402,F401
from mbrl_amd import launch  # noqa: E402

RANK = textwrap.dedent('''
    import os, sys, time
    mode, rank = sys.argv[1], int(os.environ["RANK"])
    if mode == "raise_before_collective" and rank == 1:
        raise SystemExit(7)
    if mode == "signal_before_collective" and rank == 1:
        os.abort()
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    import mbrl_amd
    from mbrl_amd import launch
    dist.init_process_group("gloo", timeout=launch.init_timeout())
    t = torch.ones(4)
    dist.all_reduce(t)
    if mode == "die_inside" and rank == 1:
        os._exit(3)                      # after the first collective, before the second
    if mode == "hang":
        time.sleep(600)
    dist.all_reduce(t)
    dist.barrier()
    print("rank %%d ok %%g" %% (rank, float(t[0])), flush=True)
''') % ROOT


def _run(mode, n=2, timeout=None, capsys=None):
    t0 = time.monotonic()
    rc = launch.launch(n, [sys.executable, "-c", RANK, mode], timeout=timeout, quiet_ranks=True,
                       env=dict(os.environ, MBRL_INIT_TIMEOUT="120"))
    return rc, time.monotonic() - t0


def test_all_ranks_fine():
    rc, dt = _run("ok")
    assert rc == 0


@pytest.mark.parametrize("mode,code", [("raise_before_collective", 7), ("signal_before_collective", 128 + 6), ("die_inside", 3)])
def test_a_dead_rank_ends_the_job_within_seconds(mode, code, capfd):
    rc, dt = _run(mode)
    assert rc == code, rc
    assert dt < 30, dt                     # not the collective's timeout (120 s here, 600 s by default)
    assert "launcher: rank 1" in capfd.readouterr().err


def test_rank_timeout_stops_every_rank():
    rc, dt = _run("hang", timeout=8)
    assert rc == 124 and dt < 40, (rc, dt)


def test_exit_code_convention():
    assert launch.exit_code(0) == 0 and launch.exit_code(5) == 5 and launch.exit_code(-6) == 134 and launch.exit_code(-9) == 137


def test_main_launcher_reports_a_failing_rank(tmp_path):
    """`python -m mbrl_amd.main --gpus 2` with a config error: both ranks raise before any collective; the launcher
    returns non-zero at once (it used to wait rank by rank and report max() of the codes)."""
    p = subprocess.run([sys.executable, "-m", "mbrl_amd.main", "--gpus", "2", "--config-file", str(tmp_path / "missing.yaml")],
                       cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT, MBRL_DIST_BACKEND="gloo"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"launcher: rank" in p.stderr
