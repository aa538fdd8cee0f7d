#!/bin/bash
# usage: tools/rocprof_stats.sh <out-dir> <title> <program> [args...]      (on the GPU box)
# rocprofv3 --kernel-trace --stats of one command → <out-dir>/stats.txt (tools/prof_summary.py table).
# The program itself goes after "--" (python3 …): no env / bash -c hop under the profiler.
set -e
out=$1; title=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw" -o p -- "$@" > "$out/run.log" 2>&1
python3 tools/prof_summary.py "$(find "$out/raw" -name '*kernel_stats.csv' | head -n 1)" "$title" > "$out/stats.txt"
rm -rf "$out/raw"
cut -c1-175 "$out/stats.txt"
