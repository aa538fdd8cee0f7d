#!/usr/bin/env python3
"""Build libhelio_diag.so (in the temporary directory, never in the tree): the same sources as libhelio.so with -DHELIO_STAMPS, i.e. with
s_memtime stamps compiled into the fused small-problem kernel (csrc/splat_fwd.hip).  A diagnostic
build only: the product library never carries a stamp (cdna_hip_programming.md §7)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doodle_amd import build as hb

import tempfile

# built on demand, OUTSIDE the tree: a diagnostic library must not travel with the product
OUT = os.path.join(tempfile.gettempdir(), "libhelio_diag.so")


def build():
    cmd = [hb.hipcc(), *hb.FLAGS, "-DHELIO_STAMPS", "-I", os.path.join(hb.ROOT, "include"), "-I", hb.CSRC, "-o", OUT,
           *[os.path.join(hb.CSRC, s) for s in hb.SOURCES]]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build())
