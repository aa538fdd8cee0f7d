#!/bin/bash
# usage: tools/pmc_pass.sh <out-dir> <counter> <program> [args...]      (on the GPU box)
# one rocprofv3 --pmc pass (counters in their own run, kernel trace only) → <out-dir>/<counter>/…_counter_collection.csv
set -e
out=$1; ctr=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out/$ctr"
rocprofv3 --pmc "$ctr" --kernel-trace --output-format csv -d "$out/$ctr" -o p -- "$@" > "$out/$ctr.log" 2>&1
ls "$out/$ctr" | head -n 3
