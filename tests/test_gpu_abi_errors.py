"""GPU: error behaviour at the C ABI (include/mbrl.h) -- return codes + mbrl_last_error instead of the reference's
asserts / exceptions (forward.py:30,33; backward.py:371,374), per-unit status bits instead of MujocoException."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eng(env="HalfCheetahEnv"):
    import mbrl_amd
    from mbrl_amd.engine import FdEngine
    return FdEngine(env)


def test_bad_arguments_return_codes():
    import torch
    import mbrl_amd
    from mbrl_amd import lib
    e = _eng()
    L, h = e.L, e.h
    z = lambda *s: torch.zeros(s, dtype=torch.float64, device="cuda")
    q, v, c = z(4, 9), z(4, 9), z(4, 6)
    p = lambda t: C.c_void_p(t.data_ptr())
    null = C.c_void_p(0)
    out = [null] * 7
    assert L.mbrl_fd_jacobians(h, 4, p(q), p(v), p(c), 0.0, 0, 1, *out, null) == -7      # eps must be > 0
    assert b"eps" in L.mbrl_last_error()
    assert L.mbrl_fd_jacobians(h, 4, p(q), p(v), p(c), 1e-6, 5, 1, *out, null) == -7     # unknown fd_mode
    assert L.mbrl_fd_jacobians(h, 4, null, p(v), p(c), 1e-6, 0, 1, *out, null) == -1      # null input
    assert L.mbrl_fd_jacobians(h, -1, p(q), p(v), p(c), 1e-6, 0, 1, *out, null) == -6     # negative N
    assert L.mbrl_fd_jacobians(None, 4, p(q), p(v), p(c), 1e-6, 0, 1, *out, null) == -1   # null model
    assert L.mbrl_fd_jacobians(h, 0, null, null, null, 1e-6, 0, 1, *out, null) == 0       # empty batch: nothing to do
    assert L.mbrl_step(h, 4, p(q), p(v), p(c), null, null, null, null, null) == -1        # outputs of a step are required
    hh = C.c_void_p()
    assert L.mbrl_model_create(b"LegEnv", 0, 0, C.byref(hh)) == -2                        # model outside the six
    assert L.mbrl_model_create(b"HalfCheetahEnv", 0, 99, C.byref(hh)) == -4               # no such device
    with pytest.raises(lib.MbrlError):
        lib.check(-7)
    with pytest.raises(ValueError):
        e.fd_jacobians(q, v, c[:3])                                                        # ragged batch
    with pytest.raises(ValueError):
        e.fd_jacobians(q, v, c, ctrl_pert=z(4, 3, 6))                                      # wrong closed-loop shape
    with pytest.raises(KeyError):
        from mbrl_amd.engine import FdEngine
        FdEngine("NoSuchEnv")


def test_all_outputs_optional_and_status_is_per_unit():
    """Any output pointer may be NULL; a unit with a non-finite state raises MBRL_ST_NONFINITE for that unit only."""
    import torch
    from mbrl_amd import lib
    e = _eng()
    rng = np.random.default_rng(0)
    N = 23                                               # not a multiple of the 5 units a wave pair holds
    q = rng.uniform(-0.1, 0.1, (N, 9)); v = 0.1 * rng.standard_normal((N, 9)); c = rng.uniform(-1, 1, (N, 6))
    q[7, 3] = np.nan
    v[12, 0] = np.inf
    full = e.fd_jacobians(q, v, c)
    st = full["status"].cpu().numpy()
    bad = np.zeros(N, bool); bad[[7, 12]] = True
    assert ((st & lib.ST_NONFINITE) != 0).tolist() == bad.tolist()
    assert torch.isfinite(full["dsds"][~torch.as_tensor(bad)]).all()
    # only dsda requested
    p = lambda t: C.c_void_p(t.data_ptr())
    qt, vt, ct = (torch.as_tensor(x, device="cuda") for x in (q, v, c))
    only = torch.full((N, 18, 6), 7.0, dtype=torch.float64, device="cuda")
    null = C.c_void_p(0)
    assert e.L.mbrl_fd_jacobians(e.h, N, p(qt), p(vt), p(ct), 1e-6, 0, 1, null, p(only), null, null, null, null, null,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    ok = ~torch.as_tensor(bad)
    assert torch.equal(only[ok], full["dsda"][ok])


def test_launch_on_a_side_stream_is_ordered():
    import torch
    e = _eng("SwimmerEnv")
    rng = np.random.default_rng(1)
    q = rng.uniform(-0.1, 0.1, (1000, 5)); v = 0.1 * rng.standard_normal((1000, 5)); c = rng.uniform(-1, 1, (1000, 2))
    ref = e.fd_jacobians(q, v, c)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = e.fd_jacobians(q, v, c)
    s.synchronize()
    for k in ref:
        assert torch.equal(ref[k], out[k]), k
