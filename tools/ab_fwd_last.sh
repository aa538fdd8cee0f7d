#!/bin/bash
# A/B of the three forms of the table kernel's consumer loop (-DHELIO_FWD_LAST_GROUP=64: every chunk whole, as before
# round 4; 8 / 16: groups of that many rays with a scalar exit between them), each built beforehand — the hipcc line of
# doodle_amd/build.py plus that -D — into build/ab/libhelio_last{0,1,2}.so.
# usage: tools/ab_fwd_last.sh OUT   (on the GPU box, from the repository root)
set -e
out=$1
cp doodle_amd/libhelio.so build/ab/libhelio_keep.so
for round in 1 2; do
  for v in 0 1 2; do
    cp build/ab/libhelio_last$v.so doodle_amd/libhelio.so
    python tools/ab_fwd_last.py "last=$v" >> "$out" 2>&1
  done
done
cp build/ab/libhelio_keep.so doodle_amd/libhelio.so
