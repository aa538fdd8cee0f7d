#!/bin/bash
# usage (on the GPU box): tools/profile_round4.sh <out-dir> [stats|pmc|all]
# The rocprofv3 evidence of round 4, every summary with ONE meaning: the dense and the culled config-4
# forward in SEPARATE runs, the backward likewise, the bench command, the few-ray forward at one shape; then the PMC passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only) that tools/pmc_traffic.py turns into profiles/r04_traffic.json.
set -e
out=$1; what=${2:-all}
mkdir -p "$out"
if [ "$what" != "pmc" ]; then
  HELIO_NOREF=1 tools/rocprof_stats.sh "$out/fwd_cfg4_dense" "config 4 forward, DENSE only: HELIO_NOREF=1 tools/bench_splat.py cfg4 512 5" python3 tools/bench_splat.py cfg4 512 5 > "$out/fwd_cfg4_dense.log"
  HELIO_NOREF=1 tools/rocprof_stats.sh "$out/fwd_cfg4_culled" "config 4 forward, CULLED only (device scratch handed over): HELIO_NOREF=1 tools/bench_splat.py cfg4 512 5c" python3 tools/bench_splat.py cfg4 512 5c > "$out/fwd_cfg4_culled.log"
  CULL=0 tools/rocprof_stats.sh "$out/bwd_cfg4_dense" "config 4 backward, DENSE: CULL=0 tools/bench_bwd_only.py cfg4 10" python3 tools/bench_bwd_only.py cfg4 10 > "$out/bwd_cfg4_dense.log"
  CULL=1 tools/rocprof_stats.sh "$out/bwd_cfg4_culled" "config 4 backward, with the lists: CULL=1 tools/bench_bwd_only.py cfg4 10" python3 tools/bench_bwd_only.py cfg4 10 > "$out/bwd_cfg4_culled.log"
  tools/rocprof_stats.sh "$out/few_512_8_512" "render_fwd_few at B=512, N=8, R=512: tools/bench_few_fwd.py 512 8 512" python3 tools/bench_few_fwd.py 512 8 512 > "$out/few_512_8_512.log"
  tools/rocprof_stats.sh "$out/bench" "python3 bench.py --no-cpu --no-extras --steps 500" python3 bench.py --no-cpu --no-extras --steps 500 > "$out/bench.log"
fi
if [ "$what" != "stats" ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    HELIO_NOREF=1 tools/pmc_pass.sh "$out/pmc_dense" $ctr python3 tools/bench_splat.py cfg4 512 5
    HELIO_NOREF=1 tools/pmc_pass.sh "$out/pmc_culled" $ctr python3 tools/bench_splat.py cfg4 512 5c
    tools/pmc_pass.sh "$out/pmc_fused" $ctr python3 bench.py --no-cpu --no-large --no-extras --steps 2000
    tools/pmc_pass.sh "$out/pmc_few8" $ctr python3 tools/bench_few_fwd.py 512 8 512
    tools/pmc_pass.sh "$out/pmc_few1" $ctr python3 tools/bench_few_fwd.py 512 1 512
  done
fi
echo profile_round4 done
