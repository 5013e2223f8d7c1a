#!/usr/bin/env python3
"""Host-side cost of one mbrl_step_record call (the trainer's rollout loop makes one per env step; the launch itself is
asynchronous).  usage: python tools/host_call_cost.py [lib.so ...]   (default: the in-tree build)
Round 2's library asked rocprim for its sort-scratch size (plan_layout) and read an environment variable on every call;
round 3 caches the size per N and reads a flag word."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

vp = C.c_void_p


def probe(path):
    L = C.CDLL(path)
    L.mbrl_model_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.mbrl_fd_plan_bytes.argtypes = [vp, C.c_int64]
    L.mbrl_fd_plan_bytes.restype = C.c_int64
    L.mbrl_step_record.argtypes = [vp, C.c_int64] + [vp] * 7 + [vp, C.c_int64, C.c_int64, C.c_int64, vp]
    L.mbrl_step.argtypes = [vp, C.c_int64] + [vp] * 7 + [vp]
    h = vp()
    assert L.mbrl_model_create(b"HalfCheetahEnv", 0, 0, C.byref(h)) == 0
    N, units = 64, 4 * 4096
    z = lambda *s: torch.zeros(s, dtype=torch.float64, device="cuda")
    q, v, c, qn, vn, r = z(N, 9), z(N, 9), z(N, 6), z(N, 9), z(N, 9), z(N)
    q[:, 1] = 0.5                                   # in the air: the cheapest possible launch
    st = torch.zeros(N, dtype=torch.int32, device="cuda")
    plan = torch.empty(int(L.mbrl_fd_plan_bytes(h, units)), dtype=torch.uint8, device="cuda")
    p = lambda t: vp(t.data_ptr())
    s = vp(torch.cuda.current_stream().cuda_stream)
    out = {}
    for name, call in (("mbrl_step", lambda: L.mbrl_step(h, N, p(q), p(v), p(c), p(qn), p(vn), p(r), p(st), s)),
                       ("mbrl_step_record", lambda: L.mbrl_step_record(h, N, p(q), p(v), p(c), p(qn), p(vn), p(r), p(st),
                                                                        p(plan), plan.numel(), units, 0, s))):
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                assert call() == 0
            best = min(best, (time.perf_counter() - t0) / 200)
            torch.cuda.synchronize()
        out[name] = best * 1e6
    print("%-40s host time per call: mbrl_step %.1f us, mbrl_step_record %.1f us" %
          (os.path.basename(path), out["mbrl_step"], out["mbrl_step_record"]), flush=True)


if __name__ == "__main__":
    libs = sys.argv[1:] or [os.path.join(ROOT, "model-based-rl_amd", "csrc", "libmbrl_hip.so")]
    if len(libs) > 1:           # one library instance per process
        import subprocess
        for l in libs:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), l])
    else:
        probe(libs[0])
