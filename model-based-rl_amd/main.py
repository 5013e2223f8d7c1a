"""Command line of the reference (main.py:79-120): --config-file, --mode, --iter, --opts; yacs-style merge and
freeze; output/<ENV>/<timestamp|NAME>/{weights,results[,recordings]} and logs.txt (main.py:12-32).
`python -m mbrl_amd.main --config-file configs/half_cheetah.yaml --opts MODEL.EPOCHS 10` (or `python mbrl_amd.py`).
Modes other than "train" (the learned-dynamics trainer, main.py:47-72) are outside the FD path."""
import argparse
import os
from datetime import datetime

from . import config, trainer


def train(cfg, iter, device=None):
    env_output_dir = os.path.join(cfg.OUTPUT.DIR, cfg.MUJOCO.ENV)
    name = "{0:%Y-%m-%d %H:%M:%S}".format(datetime.now()) if cfg.OUTPUT.NAME == "timestamp" else cfg.OUTPUT.NAME
    output_dir = os.path.join(env_output_dir, name)
    output_rec_dir = os.path.join(output_dir, "recordings")
    output_weights_dir = os.path.join(output_dir, "weights")
    output_results_dir = os.path.join(output_dir, "results")
    os.makedirs(output_dir)
    os.mkdir(output_weights_dir)
    os.mkdir(output_results_dir)
    logger = trainer.setup_logger("model.engine.trainer", output_dir, "logs")
    logger.info("Running with config:\n{}".format(cfg))
    agent = None
    for i in range(iter):
        agent = trainer.do_training(cfg, logger, output_results_dir, output_rec_dir, output_weights_dir, i, device=device)
    return agent, output_dir


def main(argv=None):
    parser = argparse.ArgumentParser(description="MI355X-native model-based RL (FD-linearised MuJoCo dynamics).")
    parser.add_argument("--config-file", default="", metavar="file", help="path to config file", type=str)
    parser.add_argument("--mode", default="train", metavar="mode", help="'train'", type=str)
    parser.add_argument("--iter", default=1, help="Number of iterations", type=int)
    parser.add_argument("--opts", help="Modify config options using the command-line", default=[], nargs=argparse.REMAINDER)
    args = parser.parse_args(argv)
    cfg = config.get_cfg_defaults()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    if args.mode != "train":
        raise SystemExit("mode %r is outside the FD path; only 'train' is built" % args.mode)
    train(cfg, args.iter)


if __name__ == "__main__":
    main()
