"""A/B on the GPU: one-launch FD (mbrl_fd_jacobians) vs plan + planned launch, same inputs; checks bits, prints times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mbrl_amd  # noqa: F401
from mbrl_amd import engine, lib
from mbrl_amd import workloads

def main():
    env = sys.argv[1] if len(sys.argv) > 1 else "HalfCheetahEnv"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 163840
    e = engine.FdEngine(env, 0, "cuda:0")
    q, v, c = workloads.s2(e, N, seed=0, horizon=40)
    for mode in (lib.FD_FORWARD, lib.FD_CENTRAL):
        a = e.fd_jacobians(q, v, c, fd_mode=mode)
        b = e.fd_jacobians(q, v, c, fd_mode=mode, planned=True)
        torch.cuda.synchronize()
        for k in a:
            same = torch.equal(a[k], b[k])
            print(env, "mode", mode, k, "bit-equal" if same else "DIFF max %.3e" % (a[k].double() - b[k].double()).abs().max().item())
        for planned in (False, True):
            for _ in range(2):
                e.fd_jacobians(q, v, c, fd_mode=mode, planned=planned, out=a)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                e.fd_jacobians(q, v, c, fd_mode=mode, planned=planned, out=a)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print(env, "mode", mode, "planned" if planned else "one-launch", "%.3f ms  %.2f M J/s" % (dt * 1e3, N / dt / 1e6))
main()
