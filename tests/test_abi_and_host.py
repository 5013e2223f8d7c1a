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
    assert L.mbrl_abi_version() == 9


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


@pytest.mark.parametrize("dims", [(2, 2, 1), (3, 3, 1), (5, 5, 2), (6, 6, 3), (9, 9, 6)])
@pytest.mark.parametrize("fd_mode", [-1, 0, 1])
def test_lane_map_covers_every_rollout_once(dims, fd_mode):
    """Host logic of the launch layout: over the 128 lanes of a wave pair every rollout of every unit
    appears exactly once (the nominal of a unit shared by the two waves: once per wave, one of them primary), every
    perturbed lane has its nominal -- and, for central differences, its -eps partner -- in its OWN wave."""
    import mbrl_amd
    from mbrl_amd import lib
    L = lib.load()
    nq, nv, nu = dims
    NP = nu + nq + nv
    R = 1 if fd_mode < 0 else (1 + 2 * NP if fd_mode == 1 else 1 + NP)
    arr = lambda: (C.c_int * 128)()
    upp, unit, roll, ref, prim = C.c_int(), arr(), arr(), arr(), arr()
    assert L.mbrl_lane_map(nq, nv, nu, fd_mode, C.byref(upp), unit, roll, ref, prim) == 0
    unit, roll, ref, prim = (list(a) for a in (unit, roll, ref, prim))
    assert upp.value >= 2 * (64 // R) and upp.value * R <= 127 + (1 if 64 % R == 0 else 0)
    seen = {}
    for i in range(128):
        if unit[i] < 0:
            continue
        w = i // 64
        assert 0 <= unit[i] < upp.value and 0 <= roll[i] < R
        # reference lane: same wave, same unit, nominal rollout
        j = w * 64 + ref[i]
        assert unit[j] == unit[i] and roll[j] == 0
        if fd_mode == 1 and roll[i] > 0 and roll[i] % 2 == 1:      # +eps lane: its -eps partner is the next lane
            assert i % 64 < 63 and unit[i + 1] == unit[i] and roll[i + 1] == roll[i] + 1
        seen.setdefault((unit[i], roll[i]), []).append((w, prim[i]))
    for u in range(upp.value):
        for p in range(R):
            hits = seen.get((u, p), [])
            if p == 0:
                assert sum(pr for _, pr in hits) == 1 and 1 <= len(hits) <= 2      # exactly one wave writes s', r
                assert len({w for w, _ in hits}) == len(hits)
            else:
                assert len(hits) == 1, (u, p, hits)
    busy = sum(1 for x in unit if x >= 0)
    if dims == (9, 9, 6) and fd_mode == 0:
        assert upp.value == 5 and busy == 126
    assert L.mbrl_lane_map(30, 30, 6, 0, None, unit_arr := arr(), arr(), arr(), arr()) == -6
