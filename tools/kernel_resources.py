"""Register / scratch use of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py splat_fwd.hip [name filter ...]

A kernel of the hot path that spills is a regression the timings show only as noise; this prints one
line per kernel: VGPRs, AGPRs, SGPRs, scratch bytes per lane, spills, waves per SIMD."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from doodle_amd import build as hb  # noqa: E402


def resources(source):
    flags = [f for f in hb.FLAGS if f not in ("-shared",)]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [hb.hipcc(), *flags, "-I", os.path.join(ROOT, "include"), "-I", hb.CSRC, "-c",
               os.path.join(hb.CSRC, source), "-o", os.path.join(tmp, "o.o"), "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out = []
    for blk in re.split(r"remark: Function Name: ", err)[1:]:
        name = blk.split()[0]
        g = lambda k: int(re.search(k + r": (\d+)", blk).group(1))  # noqa: E731
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        out.append((demangled, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                    g("VGPRs Spill"), g("SGPRs Spill"), g(r"Occupancy \[waves/SIMD\]")))
    return out


if __name__ == "__main__":
    filt = sys.argv[2:]
    print(f"{'kernel':78s} vgpr agpr sgpr scratch vspill sspill waves/simd")
    for r in resources(sys.argv[1]):
        if filt and not any(f in r[0] for f in filt):
            continue
        print(f"{r[0][:78]:78s} {r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:7d} {r[5]:6d} {r[6]:6d} {r[7]:6d}")
