"""CPU: the C-ABI library loads and exports every symbol include/mbrl.h declares (no compute calls without a
GPU), fails loudly without a GPU, and the host-side mirror of the reference's config tree behaves like yacs."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mbrl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mbrl_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import mbrl_amd
    from mbrl_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = lib.load()
    names = _declared()
    assert set(names) == set(lib.SYMBOLS)
    for n in names:
        assert getattr(L, n) is not None
    assert L.mbrl_abi_version() == 2


def test_error_paths_need_no_gpu():
    import mbrl_amd
    from mbrl_amd import lib
    L = lib.load()
    h = C.c_void_p()
    assert L.mbrl_model_create(b"NoSuchEnv", 0, 0, C.byref(h)) == -2
    assert b"unknown env" in L.mbrl_last_error()
    assert L.mbrl_model_create(None, 0, 0, C.byref(h)) == -1
    assert L.mbrl_model_dims(None, None, None, None, None) == -1
    assert L.mbrl_step(None, 1, None, None, None, None, None, None, None, None) == -1


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import mbrl_amd
    from mbrl_amd import lib
    from mbrl_amd.engine import FdEngine
    with pytest.raises(lib.MbrlError):
        FdEngine("HalfCheetahEnv")
    h = C.c_void_p()
    assert lib.load().mbrl_model_create(b"HalfCheetahEnv", 0, 0, C.byref(h)) == -3   # no HIP device, no CPU path


def test_config_tree_mirrors_reference_keys(tmp_path):
    import mbrl_amd
    from mbrl_amd import config
    cfg = config.get_cfg_defaults()
    assert cfg.MODEL.POLICY.GRAD_WEIGHTS == "average" and cfg.MODEL.NSTEPS_FOR_BACKWARD == 1
    y = tmp_path / "hc.yaml"
    y.write_text("MODEL:\n  FRAME_SKIP: 10\n  POLICY:\n    MAX_HORIZON_STEPS: 40\nMUJOCO:\n  ENV: 'HalfCheetahEnv'\n")
    cfg.merge_from_file(str(y))
    cfg.merge_from_list(["MODEL.BATCH_SIZE", "4", "SOLVER.BASE_LR", "0.01"])
    assert (cfg.MODEL.FRAME_SKIP, cfg.MODEL.BATCH_SIZE, cfg.MUJOCO.ENV) == (10, 4, "HalfCheetahEnv")
    with pytest.raises(KeyError):
        cfg.merge_from_list(["MODEL.NO_SUCH_KEY", "1"])
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.MODEL.FRAME_SKIP = 3
    assert config.cfg_for_env("Walker2dEnv").MODEL.FRAME_SKIP == 25


def test_index_is_shared_by_reference():
    """utils/index.py:1-43 semantics: the step counter captured in a snapshot is a COPY, the live one is shared."""
    import copy
    import mbrl_amd
    from mbrl_amd.agent import Index
    i = Index(0)
    alias, snap = i, copy.deepcopy(i)
    i += 1
    assert alias == 1 and snap == 0 and int(i - snap) == 1
    i.set(snap)
    assert alias == 0
