#!/bin/bash
# usage (on the GPU box): tools/runtime_knobs.sh OUT
# The driver's sample (bench.py --steps 20 --warmup 5: a 90 µs timed region = 20 launches + one fence) and the long loop under a few settings of the HIP
# runtime's own environment switches (names as the strings of libamdhip64.so give them): does how the host WAITS for the fence, or where the kernel
# arguments live, move either figure?  Three samples each, settings interleaved.
out=$1
run() {   # label, env assignments…
  label=$1; shift
  for k in 20 2000; do
    v=$(env "$@" python bench.py --steps $k --warmup 5 --no-cpu --no-large --no-extras 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$label steps=$k value=$v" >> "$out"
  done
}
for round in 1 2 3; do
  run "default                         " HELIO_KNOB=0
  run "ROC_ACTIVE_WAIT_TIMEOUT=200     " ROC_ACTIVE_WAIT_TIMEOUT=200
  run "ROC_ACTIVE_WAIT_TIMEOUT=100000  " ROC_ACTIVE_WAIT_TIMEOUT=100000
  run "HIP_FORCE_DEV_KERNARG=1         " HIP_FORCE_DEV_KERNARG=1
  run "HIP_FORCE_DEV_KERNARG=0         " HIP_FORCE_DEV_KERNARG=0
  run "AMD_DIRECT_DISPATCH=0           " AMD_DIRECT_DISPATCH=0
  run "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0" DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
  run "ROC_SKIP_KERNEL_ARG_COPY=1      " ROC_SKIP_KERNEL_ARG_COPY=1
done
