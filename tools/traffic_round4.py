#!/usr/bin/env python3
"""profiles/r04_traffic.json from the PMC passes of tools/profile_round4.sh:   traffic_round4.py <out-dir of the script> <out.json>
One pmc_traffic.py call per profiled command (each has its own pair of FETCH_SIZE / WRITE_SIZE directories), merged.
Run on the tree the profiled library was built from: every entry carries the hash of its kernel's source."""
import json, os, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
out_dir, out_json = sys.argv[1], sys.argv[2]
PASSES = [
    ("pmc_dense", ["splat_fwd_mfma_tile<4|splat_fwd_mfma_tile@N=2000,B=512,R=512|2|splat_fwd.hip"]),
    ("pmc_culled", ["splat_fwd_mfma_tile<4|splat_fwd_mfma_tile(culled)@N=2000,B=512,R=512|2|splat_fwd.hip",
                    "cull_fwd_kernel|cull_fwd_kernel@N=2000,B=512,R=512|2|cull.hip"]),
    ("pmc_fused", ["render_fwd_fused_small<4, false>|render_fwd_fused_small@N=50,B=25,R=128|1|splat_fwd.hip"]),
    ("pmc_few8", ["render_fwd_few<false, 8>|render_fwd_few@N=8,B=512,R=512|1|splat_fwd.hip"]),
    ("pmc_few1", ["render_fwd_few<false, 1>|render_fwd_few@N=1,B=512,R=512|1|splat_fwd.hip"]),
]
doc = {}
for d, specs in PASSES:
    with tempfile.NamedTemporaryFile(suffix=".json") as tmp:
        subprocess.run([sys.executable, os.path.join(HERE, "pmc_traffic.py"), os.path.join(out_dir, d, "FETCH_SIZE"),
                        os.path.join(out_dir, d, "WRITE_SIZE"), tmp.name, *specs], check=True, stdout=subprocess.DEVNULL)
        part = json.load(open(tmp.name))
    doc.update({k: v for k, v in part.items() if k != "_source" or "_source" not in doc})
doc["_source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only, KB per launch averaged over the launches of the run; "
                  "tools/profile_round4.sh → tools/traffic_round4.py → tools/pmc_traffic.py); hbm_bytes = (fetch_kb*fetch_correction + write_kb)*1024")
json.dump(doc, open(out_json, "w"), indent=1)
for k, v in doc.items():
    if k != "_source":
        print(f"{k:55s} {v['hbm_bytes'] / 1e6:9.1f} MB  ({v['source']} {v['source_sha16']})")
