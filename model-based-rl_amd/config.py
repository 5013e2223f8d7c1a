"""Config tree with the reference's keys (model/config/defaults.py:6-109) without yacs.

`CfgNode` supports what the reference's entry points use: attribute access, ``merge_from_file``
(yaml), ``merge_from_list`` (``--opts`` pairs), ``freeze`` and ``clone`` (main.py:111-114).
"""
import copy

import yaml


class CfgNode(dict):
    def __init__(self, d=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self._frozen:
            raise AttributeError("config is frozen")
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self, flag=True):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze(flag)

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError("Non-existent config key: %s%s" % (path, k))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("%s%s must be a mapping" % (path, k))
                self[k]._merge(v, path + k + ".")
            else:
                if isinstance(v, str) and isinstance(self[k], tuple):
                    v = tuple(yaml.safe_load(v.replace("(", "[").replace(")", "]")))
                self[k] = v

    def merge_from_file(self, path):
        with open(path) as fh:
            self._merge(yaml.safe_load(fh) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("Non-existent config key: %s" % key)
            node[parts[-1]] = yaml.safe_load(val) if isinstance(val, str) else val


_DEFAULTS = {
    "MODEL": {
        "META_ARCHITECTURE": "Basic", "DEVICE": "cuda", "WEIGHTS": "", "EPOCHS": 10000, "BATCH_SIZE": 8,
        "POLICY": {
            "ARCH": "StochasticPolicy", "MAX_HORIZON_STEPS": 100, "LAYERS": [64, 64, 32, 8], "STD_SCALER": 1e-1,
            "SOFT_LOWER_STD_BOUND": 1e-4, "SOFT_LOWER_STD_THRESHOLD": 1e-1, "OBS_SCALER": False,
            "FORGET_COUNT_OBS_SCALER": 5000, "GAMMA": 0.99, "METHOD": "None", "INITIAL_LOG_SD": 0.0,
            "INITIAL_SD": 0.0, "INITIAL_ACTION_MEAN": 0.0, "INITIAL_ACTION_SD": 0.1, "GRAD_WEIGHTS": "average",
            "NETWORK": False,
        },
        "NSTEPS_FOR_BACKWARD": 1, "FRAME_SKIP": 1, "TIMESTEP": 0.0, "RANDOM_SEED": 0,
        # additions of this build (not in the reference): FD scheme selection
        "FD_EPS": 1e-6, "FD_MODE": "forward",
        # "fused": H step launches + ONE FD launch over B*H units + one scan per epoch; "ops": the reference's op graph
        "ENGINE_PATH": "fused", "FD_OVERLAP_STEPS": 4,
    },
    "MUJOCO": {"ENV": "InvertedPendulumEnv", "ASSETS_PATH": "./mujoco/assets/", "REWARD_SCALE": 1,
               "CLIP_ACTIONS": True, "POOL_SIZE": {}},
    "EXPERIENCE_REPLAY": {"SIZE": 2 ** 15, "SHUFFLE": True, "ENV_INIT_STATE_NUM": 2 ** 15 * 3 / 4},
    "SOLVER": {"OPTIMIZER": "adam", "BASE_LR": 0.001, "STD_LR_FACTOR": 0.001, "BIAS_LR_FACTOR": 2,
               "WEIGHT_DECAY_SD": 0.0, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0005, "WEIGHT_DECAY_BIAS": 0,
               "ADAM_BETAS": (0.9, 0.999)},
    "OUTPUT": {"DIR": "./output", "NAME": "timestamp"},
    "LOG": {"PERIOD": 1, "PLOT": {"ENABLED": True, "DISPLAY_PORT": 8097, "ITER_PERIOD": 1},
            "TESTING": {"ENABLED": True, "ITER_PERIOD": 1, "RECORD_VIDEO": False, "COUNT_PER_ITER": 1},
            "CHECKPOINT_PERIOD": 25000},
}

# per-env overrides of configs/<env>.yaml that the hot path reads (FRAME_SKIP :7, BATCH_SIZE :6,
# MAX_HORIZON_STEPS :14)
ENV_YAML = {
    "InvertedPendulumEnv": dict(FRAME_SKIP=2, BATCH_SIZE=8, MAX_HORIZON_STEPS=50),
    "InvertedDoublePendulumEnv": dict(FRAME_SKIP=4, BATCH_SIZE=8, MAX_HORIZON_STEPS=50),
    "SwimmerEnv": dict(FRAME_SKIP=10, BATCH_SIZE=1, MAX_HORIZON_STEPS=40),
    "HopperEnv": dict(FRAME_SKIP=20, BATCH_SIZE=1, MAX_HORIZON_STEPS=40),
    "Walker2dEnv": dict(FRAME_SKIP=25, BATCH_SIZE=1, MAX_HORIZON_STEPS=40),
    "HalfCheetahEnv": dict(FRAME_SKIP=10, BATCH_SIZE=1, MAX_HORIZON_STEPS=40),
}


def get_cfg_defaults():
    """Same contract as model/config/defaults.py:105-109 (a fresh clone of the defaults)."""
    return CfgNode(copy.deepcopy(_DEFAULTS))


def cfg_for_env(env, **policy_over):
    """Defaults + the values of configs/<env>.yaml that the FD path reads."""
    cfg = get_cfg_defaults()
    y = ENV_YAML[env]
    cfg.MUJOCO.ENV = env
    cfg.MODEL.FRAME_SKIP = y["FRAME_SKIP"]
    cfg.MODEL.BATCH_SIZE = y["BATCH_SIZE"]
    cfg.MODEL.POLICY.MAX_HORIZON_STEPS = y["MAX_HORIZON_STEPS"]
    for k, v in policy_over.items():
        cfg.MODEL.POLICY[k] = v
    return cfg
