"""Command line of the reference (main.py:79-120): --config-file, --mode, --iter, --opts; yacs-style merge and
freeze; output/<ENV>/<timestamp|NAME>/{weights,results[,recordings]} and logs.txt (main.py:12-32).
`python -m mbrl_amd.main --config-file configs/half_cheetah.yaml --opts MODEL.EPOCHS 10` (or `python mbrl_amd.py`).
Modes other than "train" (the learned-dynamics trainer, main.py:47-72) are outside the FD path."""
import argparse
import os
import sys
from datetime import datetime

from . import launch


def train(cfg, iter, device=None, engine=None):
    """main.py:12-44.  Under torch.distributed (one process per GPU) rank 0 creates the output directories and
    broadcasts their name (the default name is a timestamp); the other ranks log to stdout only."""
    import torch.distributed as dist
    from . import trainer
    multi = dist.is_available() and dist.is_initialized()     # also a forced group of one rank (MBRL_FORCE_DIST)
    rank = dist.get_rank() if multi else 0
    env_output_dir = os.path.join(cfg.OUTPUT.DIR, cfg.MUJOCO.ENV)
    name = "{0:%Y-%m-%d %H:%M:%S}".format(datetime.now()) if cfg.OUTPUT.NAME == "timestamp" else cfg.OUTPUT.NAME
    if multi:
        box = [name]
        dist.broadcast_object_list(box, src=0)
        name = box[0]
    output_dir = os.path.join(env_output_dir, name)
    output_rec_dir = os.path.join(output_dir, "recordings")
    output_weights_dir = os.path.join(output_dir, "weights")
    output_results_dir = os.path.join(output_dir, "results")
    if rank == 0:
        os.makedirs(output_dir)
        os.mkdir(output_weights_dir)
        os.mkdir(output_results_dir)
    if multi:
        dist.barrier()
    logger = trainer.setup_logger("model.engine.trainer", output_dir if rank == 0 else None, "logs")
    if rank == 0:
        logger.info("Running with config:\n{}".format(cfg))
    agent = None
    for i in range(iter):
        agent = trainer.do_training(cfg, logger, output_results_dir, output_rec_dir, output_weights_dir, i, device=device,
                                    engine=engine)
    return agent, output_dir


def launch_ranks(n, argv, timeout=None):
    """`--gpus n`: start n ranks of this command line as fresh child processes, one per GPU, and supervise them
    (launch.py).  The parent has not touched the GPU and never does (nothing is exec'ed after HIP initialisation).  The
    first rank that fails ends the job: the others are stopped instead of waiting in a collective, and its exit code is
    returned (128 + signal for a signal death)."""
    return launch.launch(n, [sys.executable, "-m", "mbrl_amd.main"] + list(argv), timeout=timeout, quiet_ranks=False)


def init_distributed(backend=None):
    """One process per GPU, started by `--gpus n` or by torch.distributed.run: join the process group named by RANK /
    WORLD_SIZE / MASTER_* and bind this rank to GPU LOCAL_RANK.  Returns the device string (None: single process).
    MBRL_FORCE_DIST=1 joins a process group with ONE rank too (the RCCL branch -- init, barrier, all-reduce on device
    tensors -- then runs on a one-GPU box exactly as it does on eight)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("MBRL_FORCE_DIST", "") not in ("", "0")
    if world <= 1 and not force:
        return None
    if "MASTER_ADDR" not in os.environ:     # forced single rank without a launcher
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(launch.free_port()), RANK="0", WORLD_SIZE="1")
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    if backend is None:     # MBRL_DIST_BACKEND=gloo: ranks that share one GPU (tests on a one-GPU box); RCCL needs a GPU per rank
        backend = os.environ.get("MBRL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=launch.init_timeout())
        return "cuda:%d" % local
    dist.init_process_group(backend, timeout=launch.init_timeout())
    if torch.cuda.is_available():
        local %= max(torch.cuda.device_count(), 1)      # rehearsal: more ranks than GPUs share them (as bench.py does)
        torch.cuda.set_device(local)
        return "cuda:%d" % local
    return None


def main(argv=None):
    parser = argparse.ArgumentParser(description="MI355X-native model-based RL (FD-linearised MuJoCo dynamics).")
    parser.add_argument("--config-file", default="", metavar="file", help="path to config file", type=str)
    parser.add_argument("--mode", default="train", metavar="mode", help="'train'", type=str)
    parser.add_argument("--iter", default=1, help="Number of iterations", type=int)
    parser.add_argument("--gpus", default=1, type=int,
                        help="GPUs of this node to shard the episodes over (one process each, RCCL all-reduce)")
    parser.add_argument("--rank-timeout", default=None, type=float,
                        help="--gpus N: stop every rank after this many seconds (default: MBRL_RANK_TIMEOUT or none)")
    parser.add_argument("--opts", help="Modify config options using the command-line", default=[], nargs=argparse.REMAINDER)
    args = parser.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv, args.rank_timeout))
    from . import config
    device = init_distributed()
    cfg = config.get_cfg_defaults()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    if args.mode != "train":
        raise SystemExit("mode %r is outside the FD path; only 'train' is built" % args.mode)
    train(cfg, args.iter, device=device)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
