#!/usr/bin/env python3
"""Condense a rocprofv3 ``*_kernel_stats.csv`` into a short table (kernel names cut to
90 characters) for profiles/.   usage: prof_summary.py <kernel_stats.csv> [title]"""
import csv
import sys


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    rows = list(csv.DictReader(open(path)))
    print(f"# {title}")
    print("# source: rocprofv3 --kernel-trace --stats (durations in microseconds)")
    print(f"{'kernel':92s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows:
        name = r["Name"].replace("void ", "")
        name = name if len(name) <= 90 else name[:87] + "..."
        print(f"{name:92s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} "
              f"{float(r['MaxNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main()
