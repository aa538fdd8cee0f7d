"""Ahead-of-time build of libhelio.so (hipcc, gfx950 only).

The library is built IN-TREE (doodle_amd/libhelio.so) so that it travels with the
repository snapshot to the GPU box; nothing is JIT-compiled at import time.
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhelio.so")
COMM_LIB = os.path.join(HERE, "libhelio_comm.so")      # RCCL wrapper, separate so that the
COMM_SOURCES = ["comm.hip"]                             # kernels library has no RCCL dependency
SOURCES = ["abi.hip", "geometry.hip", "splat_fwd.hip", "splat_bwd.hip", "step_losses.hip", "edt.hip", "cull.hip"]
HEADERS = [os.path.join(CSRC, "helio_math.h"), os.path.join(CSRC, "ray_trace.h"), os.path.join(CSRC, "step_loss_math.h"),
           os.path.join(CSRC, "cull.h"), os.path.join(CSRC, "cull_math.h"), os.path.join(CSRC, "geometry_bwd_ray.h"),
           os.path.join(ROOT, "include", "helio.h"), os.path.join(ROOT, "include", "helio_comm.h")]
# -ffp-contract=off: the geometry stage is bit-faithful to the reference's fp32 CPU
# arithmetic; FMAs appear only where written (helio_math.h).  Division and sqrt stay
# correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
# -amdgpu-kernarg-preload-count=14: the first 14 dwords of every kernel's arguments arrive in scalar
# registers with the wave launch instead of through a load from the kernel-argument segment (≈0.8 µs
# for the first touch in a freshly launched kernel, which is what the launch-bound small kernels wait
# for first); kernels keep the compatibility prologue for firmware that does not preload.
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-pass-failed",
         "-mllvm", "-amdgpu-kernarg-preload-count=14"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def is_stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(COMM_LIB):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(COMM_LIB))
    deps = [os.path.join(CSRC, s) for s in SOURCES + COMM_SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), *FLAGS, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB,
           *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib")
    cmd = [hipcc(), "-O2", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
           "-I", os.path.join(ROOT, "include"), "-o", COMM_LIB,
           *[os.path.join(CSRC, s) for s in COMM_SOURCES], "-L", rocm_lib, "-lrccl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


def hostbind_path():
    import glob
    hits = glob.glob(os.path.join(HERE, "_hostbind*.so"))
    return hits[0] if hits else None


def build_hostbind(force: bool = False, verbose: bool = False):
    """The compiled torch binding of the C ABI (csrc/hostbind.cpp → doodle_amd/_hostbind*.so).
    Host C++ against the torch headers (≈1 min); links libhelio.so, so build() runs first."""
    src = os.path.join(CSRC, "hostbind.cpp")
    so = hostbind_path()
    if not force and so and os.path.getmtime(so) >= max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "helio.h"))):
        return so
    build(force=False, verbose=verbose)
    cmd = [sys.executable, os.path.join(HERE, "setup_hostbind.py"), "-q", "build_ext", "--inplace",
           "--build-temp", os.path.join(ROOT, "build", "hostbind")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=ROOT)
    return hostbind_path()


def audit() -> None:
    """Static audit of the compiled kernels the build depends on the compiler for (tools/audit_fused_late.py: the
    hand-issued kernel-argument loads of the fused small-problem kernel).  Raises on a violation."""
    tool = os.path.join(ROOT, "tools", "audit_fused_late.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("kernel audit failed — this hipcc does not produce a safe libhelio.so:\n" + out.stdout + out.stderr)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_hostbind(force="--force" in sys.argv, verbose=True))
    if "--audit" in sys.argv:
        audit()
        print("kernel audit clean")
