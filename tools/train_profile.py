"""cProfile of 10 steady-state training epochs (HalfCheetah, B = 4096): where the HOST time goes."""
import os, sys, tempfile, logging, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import config, trainer

def run(epochs):
    cfg = config.get_cfg_defaults()
    cfg.merge_from_file(os.path.join(os.path.dirname(__file__), "..", "configs", "half_cheetah.yaml"))
    cfg.merge_from_list(["MODEL.BATCH_SIZE", "4096", "MODEL.EPOCHS", str(epochs), "LOG.PERIOD", "1000"])
    cfg.freeze()
    d = tempfile.mkdtemp()
    lg = logging.getLogger("q%d" % epochs); lg.setLevel(logging.ERROR)
    trainer.do_training(cfg, lg, d, d, d, 1)
    torch.cuda.synchronize()

run(3)
pr = cProfile.Profile(); pr.enable(); run(10); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
