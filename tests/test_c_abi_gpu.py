"""The C ABI used WITHOUT Python or torch: a small HIP host program (tests/c/abi_smoke.cpp) links
libhelio.so and the C oracle, renders through helio_render_fwd with raw hipMalloc'd buffers and
compares in-process.  Needs g++, the ROCm headers and an MI355X."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_client_of_the_abi(tmp_path):
    obj, exe = str(tmp_path / "oracle.o"), str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-c",
                           os.path.join(ROOT, "oracle", "helio_oracle.c"), "-o", obj])
    lib_dir = os.path.join(ROOT, "doodle_amd")
    # host-only program: g++ against the HIP runtime headers, linked with libamdhip64 and libhelio
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "c", "abi_smoke.cpp"), obj, "-I", os.path.join(ROOT, "include"),
                           "-L", lib_dir, "-lhelio", "-L/opt/rocm/lib", "-lamdhip64",
                           f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    for size in (("6", "37", "96"), ("3", "130", "260"), ("40", "300", "256")):
        out = subprocess.run([exe, *size], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "ABI SMOKE OK" in out.stdout, out.stdout + out.stderr
