"""Build the HIP extension in-tree: csrc/models_gen.hpp (generated) + csrc/libmbrl_hip.so (hipcc, gfx950)."""
import os
import shutil
import subprocess

from . import compile_planar

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmbrl_hip.so")
SOURCES = ["mbrl_hip.hip", "planar_lean.hpp", "wave_or.hpp", "models_gen.hpp"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "modelspec.py"),
                                                       os.path.join(HERE, "compile_planar.py"),
                                                       os.path.abspath(__file__),       # the compiler flags live here
                                                       os.path.join(HERE, "..", "include", "mbrl.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# -pragma-unroll-threshold: the tree-sparse factorisation must unroll completely (register arrays);
# -disable-machine-licm: loop-invariant integer bookkeeping must not be hoisted into 100+ registers;
# no-NaN / no-infinity / no-signed-zero arithmetic: lets x * 0.0 and x + 0.0 fold (model tables are full of zeros)
# and drops the canonicalisation before min / max; non-finite results are still caught, by a test on the bits
# (t_finite in planar_lean.hpp) that these flags cannot fold away.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed",
               "-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-disable-machine-licm",
               "-mllvm", "-amdgpu-use-amdgpu-trackers=1",   # GCN register-pressure trackers in the scheduler: ~30 % fewer AGPR copies
               "-mllvm", "-amdgpu-sched-strategy=max-ilp",  # occupancy is pinned at one wave per SIMD anyway: schedule for
                                                            # latency, not for registers (-2.7 %; iterative-ilp, minreg slower)
               "-fno-honor-nans", "-fno-honor-infinities", "-fno-signed-zeros"]


def compile_lib(out, extra_flags=()):
    """One hipcc line: csrc/mbrl_hip.hip -> `out` (also used by the profiling tools for their variants)."""
    cmd = [_hipcc()] + HIPCC_FLAGS + list(extra_flags) + ["-o", out + ".tmp", os.path.join(CSRC, "mbrl_hip.hip")]
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(out + ".tmp", out)
    return out


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libmbrl_hip.so (cross-compiles without a GPU)."""
    compile_planar.write_header()
    if not force and not needs_build():
        return LIB
    return compile_lib(LIB, ["-Rpass-analysis=kernel-resource-usage"] if verbose else [])
