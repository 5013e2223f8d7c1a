"""Self-launcher of the multi-GPU entry points (`bench.py --gpus N`, `python -m mbrl_amd.main --gpus N`): one fresh
child process per rank, supervised.  Imports nothing that touches the GPU (no torch): the parent never initialises HIP
and nothing is ever exec'ed from a process that has.

The reference is single-process (model/engine/trainer.py:77); this has no counterpart there.

Why supervised: the ranks meet in collectives (rendezvous, the gradient all-reduce, barriers).  If one of them dies --
`--gpus` larger than the device count, a GPU fault, a config error -- its siblings would sit in the next collective until
somebody's time limit.  So the parent polls: the first rank that exits non-zero (or is killed by a signal) ends the job;
the others are terminated (SIGTERM, then SIGKILL after a grace period -- they are fresh children of this parent, never
re-exec'ed) and the launcher returns that rank's code, 128 + signal for a signal death, with the rank's id on stderr.
"""
import os
import signal
import socket
import subprocess
import sys
import time

RANK_TIMEOUT_ENV = "MBRL_RANK_TIMEOUT"      # seconds for the whole job (0 / unset: no limit); --rank-timeout sets it
INIT_TIMEOUT_ENV = "MBRL_INIT_TIMEOUT"      # seconds for init_process_group and every collective (default 600)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return env


def exit_code(returncode):
    """Shell convention: a child killed by signal s reports 128 + s (Popen gives -s), so a signal death never reads
    as success next to a sibling's 0."""
    return 128 - returncode if returncode < 0 else returncode


def _stop(procs, grace):
    for p in procs:
        if p.poll() is None:
            p.terminate()
    deadline = time.monotonic() + grace
    for p in procs:
        while p.poll() is None and time.monotonic() < deadline:
            time.sleep(0.05)
        if p.poll() is None:
            p.kill()
    for p in procs:
        p.wait()


def supervise(procs, timeout=None, grace=5.0, poll=0.1, err=None):
    """Wait for the ranks `procs` (Popen objects, index = rank).  Returns 0 when all exit 0.  On the first non-zero
    exit, or when `timeout` seconds have passed, stops the others and returns the failing rank's code (124 for the
    time limit).  Never waits on one particular child while another may already be dead."""
    err = err or sys.stderr
    t0 = time.monotonic()
    try:
        while True:
            alive = False
            for r, p in enumerate(procs):
                rc = p.poll()
                if rc is None:
                    alive = True
                elif rc != 0:
                    code = exit_code(rc)
                    how = "was killed by signal %d" % -rc if rc < 0 else "exited with code %d" % rc
                    print("launcher: rank %d %s; stopping the other %d rank(s)" % (r, how, len(procs) - 1), file=err, flush=True)
                    _stop(procs, grace)
                    return code
            if not alive:
                return 0
            if timeout and time.monotonic() - t0 > timeout:
                print("launcher: rank time limit of %.0f s reached; stopping all %d rank(s)" % (timeout, len(procs)),
                      file=err, flush=True)
                _stop(procs, grace)
                return 124
            time.sleep(poll)
    except BaseException:       # Ctrl-C in the parent, ...: never leave ranks behind
        _stop(procs, grace)
        raise


def launch(n, cmd, timeout=None, quiet_ranks=True, env=None):
    """Start `cmd` (argv list) n times as ranks 0..n-1 of one job on 127.0.0.1 and supervise them.  Rank 0's stdout is
    the launcher's; the other ranks' stdout is dropped when quiet_ranks (their stderr is always shown)."""
    if timeout is None:
        timeout = float(os.environ.get(RANK_TIMEOUT_ENV, "0") or 0) or None
    port = free_port()
    procs = []
    try:
        for r in range(n):
            procs.append(subprocess.Popen(cmd, env=rank_env(r, n, port, env),
                                          stdout=subprocess.DEVNULL if (quiet_ranks and r) else None))
    except BaseException:
        _stop(procs, 1.0)
        raise
    return supervise(procs, timeout=timeout)


def init_timeout():
    """timedelta for init_process_group(timeout=...): a rank whose peers never arrive gives up instead of waiting for ever."""
    from datetime import timedelta
    return timedelta(seconds=float(os.environ.get(INIT_TIMEOUT_ENV, "600") or 600))
