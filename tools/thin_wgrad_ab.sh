#!/bin/bash
# the thin-input weight gradients (wgrad_thinx_kernel) against the kernels they replace: correctness on small ragged shapes (CHECK=1: torch's weight gradient of the
# stored operands) and time at the step's shapes.  HV_WGRAD_THIN: 1 = the 5x5 stems on wgrad_thinx_kernel
for t in 0 1; do
  echo "HV_WGRAD_THIN=$t"
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 2 40 48 4 16 5 1 2 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 3 37 29 4 8 5 1 2 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 3 44 36 4 64 4 2 1 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t python tools/bench_wgrad.py 16 256 256 4 16 5 1 2 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t python tools/bench_wgrad.py 32 256 256 4 64 4 2 1 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t python tools/bench_wgrad.py 16 256 256 4 64 4 2 1 2>&1 | grep "^W "
done
for w in 512 2048; do echo "WGS=$w"; HV_WGRAD_THIN_WGS=$w python tools/bench_wgrad.py 16 256 256 4 16 5 1 2 2>&1 | grep "^W "; HV_WGRAD_THIN_WGS=$w python tools/bench_wgrad.py 32 256 256 4 64 4 2 1 2>&1 | grep "^W "; done
