"""Static audits of the hand-written kernels (CPU only: hipcc cross-compiles gfx950 to assembly).

tools/audit_fused_late.py: the fused small-problem forward kernel issues the loads of its late kernel
arguments itself (inline asm) and waits for them after half of the trace; between issue and wait the
compiler must neither touch the destination registers nor emit scalar-memory / LDS traffic of its
own (cdna_hip_programming.md §5.7), and the product library must carry no s_memtime stamp."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_late_argument_loads_of_the_fused_kernel_are_left_alone_by_the_compiler():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_fused_late.py")], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "audited 3 kernels: clean" in out.stdout, out.stdout + out.stderr


def test_product_sources_keep_their_stamps_behind_the_diagnostic_macro():
    import re
    src = open(os.path.join(ROOT, "doodle_amd", "csrc", "splat_fwd.hip")).read()
    code = re.sub(r"//[^\n]*", "", src)                  # comments may talk about stamps
    first = code.index("#ifdef HELIO_STAMPS")
    defs_end = code.index("#else", first)
    outside = code[:first] + code[defs_end:]
    assert "s_memtime" not in outside and "s_memrealtime" not in outside
    from doodle_amd import build as hb
    assert "-DHELIO_STAMPS" not in hb.FLAGS


def test_every_environment_switch_the_product_reads_is_in_the_readme():
    """HELIO_* variables read with getenv / os.environ in the product are A/B switches a reader must be able
    to find: each one appears in README.md's table."""
    import glob
    import re
    names = set()
    for path in glob.glob(os.path.join(ROOT, "doodle_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "doodle_amd", "*.py")) \
            + [os.path.join(ROOT, "bench.py")]:
        text = open(path, errors="ignore").read()
        names.update(re.findall(r'getenv\("(HELIO_[A-Z0-9_]+)"\)', text))
        names.update(re.findall(r'environ(?:\.get)?[\(\[]"(HELIO_[A-Z0-9_]+)"', text))
    readme = open(os.path.join(ROOT, "README.md")).read()
    missing = sorted(n for n in names if n not in readme)
    assert len(names) >= 10 and not missing, missing
