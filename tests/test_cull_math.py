"""CPU: the arithmetic of the culling criterion (doodle_amd/csrc/cull_math.h — the header the HIP library compiles)
checked by brute force on the host, bit for bit: exponent_floor() is a lower bound of every exponent the footprint
kernels compute over a tile, for ascending, descending and shuffled coordinate arrays; NaN and plane-parallel rays
are always kept (tests/c/cull_floor.cpp).  What the GPU adds on top — v_exp_f32 and the MFMA accumulate — is covered
by the bit-equality tests of tests/test_cull_gpu.py."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_exponent_floor_is_a_lower_bound_of_every_computed_exponent(tmp_path):
    exe = str(tmp_path / "cull_floor")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "doodle_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "cull_floor.cpp"), "-o", exe])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "CULL FLOOR OK" in run.stdout, run.stdout[-3000:] + run.stderr[-1000:]
