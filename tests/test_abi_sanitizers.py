""""compile-time -fsanitize=address host build of the C-ABI layer".

The product's host layer — doodle_amd/csrc/abi.hip, the extern "C" entry points of include/helio.h
with their argument validation — is compiled as host C++ with AddressSanitizer + UBSan and driven
by tests/c/abi_san.cpp WITHOUT a device: every helio::launch_* it would call is replaced by a
counting stub generated here from abi.hip's own forward declarations, so the run proves that
invalid arguments (null pointers, bad sizes, misaligned buffers, bad strides, reserved tickets)
return HELIO_E_INVALID before any launch, and that well-formed calls get through.  CPU only
(GPU sanitizers are not available on the pool)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

ABI = os.path.join(ROOT, "doodle_amd", "csrc", "abi.hip")
HIP_INC = "/opt/rocm/include"
HIP_LIB = "/opt/rocm/lib"


def launch_stubs() -> str:
    """`namespace helio { … }` forward declarations at the top of abi.hip → counting definitions."""
    src = open(ABI).read()
    block = src[src.index("namespace helio {") + len("namespace helio {"):src.index("}  // namespace helio")]
    out = ['#include <hip/hip_runtime.h>', '#include "helio.h"', "int g_launches = 0;", "namespace helio {"]
    n = 0
    for decl in block.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"^(void|int|bool|long) (\w+)\((.*)\)$", decl)
        assert m, decl
        ret, name = m.group(1), m.group(2)
        body = "++g_launches;" if name.startswith("launch_") else ""
        # the size rules the host layer consults are part of what is under test: restate them
        rules = {
            "splat_bwd_blocks": "return (a0 + 63) / 64;",
            "render_is_fused": "{ const long t = (long)a0 * ((a2 + 127) / 128) * ((a2 + 127) / 128); const long nb = (a2 + 31) / 32; return a1 <= 8 || (a1 <= 256 && t < 512 && (long)a0 * a1 * nb * nb <= 225000); }",
            "render_is_fused_plain": "{ const long t = (long)a0 * ((a2 + 127) / 128) * ((a2 + 127) / 128); const long nb = (a2 + 31) / 32; return (a1 <= 8 && (long)a0 * a2 * a2 >= (1l << 21)) || (a1 <= 256 && t < 512 && (long)a0 * a1 * nb * nb <= 225000 && (long)a0 * nb * nb <= 4096); }",
            "step_losses_chunks": "return 4;",
            "step_losses_ray_wgs": "return 4;",
            "step_losses_max_mask_batch": "return 4096;",
            "splat_bwd_is_few": "return a1 <= 8 || (a1 <= 16 && a0 <= 64) || (a1 <= 32 && a0 <= 8);",
            "env_step_fused_workspace": "return 1024;",
            "render_fwd_choice": "return a1 <= 256 ? 12 : 6;",
            "render_bwd_choice": "return a1 <= 8 ? 4 : 10;",
            "splat_fwd_scratch_bytes": "return (a1 >= 192 && (a3 == 5 || a3 == 0)) ? 4096 : 0;",
            "splat_fwd_scratch_required": "return (a3 >= 14 && a3 <= 17) ? 8192 : 0;",
            "splat_bwd_scratch_bytes": "return (a1 > 256 && (a3 == 2 || a3 == 0)) ? 4096 : 0;",
            "render_bwd_is_fused": "{ if (a1 > 32 && a1 <= 192 && a2 > 64 && 2l * ((a2 + (a2 <= 128 ? 127 : 255)) / (a2 <= 128 ? 128 : 256)) * ((a1 + 63) / 64) * a0 >= (a2 <= 128 ? 800 : 240)) return false; const bool few = a1 <= 8 && (a1 <= 2 || (long)a0 * a2 * a2 >= (1l << 21)); const long wg = (long)a0 * ((a1 + 31) / 32); if (few || a2 > 256) return false; if (a2 <= 128 && wg <= 64) return true; if (a2 > 64 && a2 <= 128) return wg >= 160 && (a1 <= 64 || wg <= 512); if (a2 > 128) return a1 <= 64 && wg >= 256; return false; }",
        }
        args = [a.strip() for a in m.group(3).split(",")] if m.group(3).strip() else []
        named = ", ".join(f"{a} a{k}" for k, a in enumerate(args))
        if name in rules:
            out.append(f"{ret} {name}({named}) {{ {rules[name]} }}")
        else:
            assert name.startswith("launch_"), f"no stub rule for helio::{name}"
            tail = {"void": "", "bool": " return true;"}.get(ret, " return 0;")
            out.append(f"{ret} {name}({named}) {{ {body}{tail} }}")
        n += 1
    assert n >= 20
    out.append("}  // namespace helio")
    return "\n".join(out) + "\n"


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists(os.path.join(HIP_INC, "hip", "hip_runtime.h")),
                    reason="needs g++ and the ROCm headers")
def test_abi_argument_validation_under_address_and_ub_sanitizers(tmp_path):
    stubs = tmp_path / "abi_san_stubs.cpp"
    stubs.write_text(launch_stubs())
    exe = str(tmp_path / "abi_san")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-D__HIP_PLATFORM_AMD__", "-I", HIP_INC, "-I", os.path.join(ROOT, "include"),
           "-x", "c++", ABI, str(stubs), os.path.join(ROOT, "tests", "c", "abi_san.cpp"),
           "-L", HIP_LIB, "-lamdhip64", f"-Wl,-rpath,{HIP_LIB}", "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and "ABI SAN OK" in run.stdout, (run.stdout + run.stderr)[-4000:]
