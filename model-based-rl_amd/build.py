"""Build the HIP extension in-tree: csrc/models_gen.hpp (generated) + csrc/libmbrl_hip.so (hipcc, gfx950)."""
import os
import shutil
import subprocess

from . import compile_planar

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmbrl_hip.so")
SOURCES = ["mbrl_hip.hip", "planar_lean.hpp", "models_gen.hpp"]


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
                                                       os.path.join(HERE, "..", "include", "mbrl.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libmbrl_hip.so (cross-compiles without a GPU)."""
    compile_planar.write_header()
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed", "-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-disable-machine-licm",
           "-o", LIB + ".tmp", os.path.join(CSRC, "mbrl_hip.hip")]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB
