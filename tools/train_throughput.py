"""End-to-end trainer throughput on one GPU: wall time per epoch of do_training for HalfCheetah (CONFIG=<name of a file in
configs/> for another model; VariationalOptimization, open loop, the configuration's horizon) at several batch sizes.  An epoch = B episodes x 40 env steps forward + the linearisation of all
B*40 (s, a) + the backward scan + the optimiser step."""
import os, sys, tempfile, time, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import config, trainer

def run(B, epochs):
    cfg = config.get_cfg_defaults()
    cfg.merge_from_file(os.path.join(os.path.dirname(__file__), "..", "configs", os.environ.get("CONFIG", "half_cheetah") + ".yaml"))
    cfg.merge_from_list(["MODEL.BATCH_SIZE", str(B), "MODEL.EPOCHS", str(epochs), "LOG.PERIOD", "1000",
                         "MODEL.FD_OVERLAP_STEPS", os.environ.get("OVERLAP", "4")])      # 4 = the configuration default
    cfg.freeze()
    d = tempfile.mkdtemp()
    lg = logging.getLogger("quiet%d_%d" % (B, epochs)); lg.setLevel(logging.ERROR)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trainer.do_training(cfg, lg, d, d, d, 1)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


_c = config.get_cfg_defaults(); _c.merge_from_file(os.path.join(os.path.dirname(__file__), "..", "configs", os.environ.get("CONFIG", "half_cheetah") + ".yaml"))
H = int(_c.MODEL.POLICY.MAX_HORIZON_STEPS)
run(64, 2)     # load the library, warm the allocator
for B in [int(x) for x in os.environ.get("BATCHES", "1024,4096,16384").split(",")]:
    base = min(run(B, 3) for _ in range(2))
    per = (min(run(B, 13) for _ in range(2)) - base) / 10
    print("B %6d: %.1f ms per epoch, %.2f M env steps/s = linearised (s, a)/s" % (B, per * 1e3, B * H / per / 1e6), flush=True)
