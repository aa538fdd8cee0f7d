#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs of
the same command) → the JSON bench.py reads for `roofline.traffic`.

    pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<kernel-name substring>|<key>|<fetch correction>|<source .hip>" ...

Each entry is stamped with the sha256 (16 hex digits) of the kernel's source file AS IT IS NOW: run
this on the same tree the profiled library was built from.  bench.py refuses an entry whose hash does
not match the source the library is built from.  fetch correction: 2 when every global read of the
kernel is a 16-byte-per-lane load (gfx950's FETCH_SIZE counts those at one half,
MI355X_MICROARCH.md §HBM), 1 = as read (uncalibrated)."""
import collections, csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def averages(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = averages(fetch_dir, "FETCH_SIZE"), averages(write_dir, "WRITE_SIZE")
    doc = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KB per launch, averaged over the "
                      "launches of the run); hbm_bytes = (fetch_kb*fetch_correction + write_kb)*1024; tools/pmc_traffic.py"}
    for spec in sys.argv[4:]:
        sub, key, corr, src = spec.split("|")
        fk = [(k, v) for k, v in fetch.items() if sub in k]
        wk = [(k, v) for k, v in write.items() if sub in k]
        assert len(fk) == 1 and len(wk) == 1, (sub, [k for k, _ in fk], [k for k, _ in wk])
        (f_kb, f_n), (w_kb, w_n) = fk[0][1], wk[0][1]
        sha = hashlib.sha256(open(os.path.join(ROOT, "doodle_amd", "csrc", src), "rb").read()).hexdigest()[:16]
        doc[key] = {"kernel": fk[0][0][:120], "fetch_kb": round(f_kb, 1), "write_kb": round(w_kb, 1), "launches": [f_n, w_n],
                    "fetch_correction": float(corr), "hbm_bytes": int((f_kb * float(corr) + w_kb) * 1024),
                    "source": src, "source_sha16": sha}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
