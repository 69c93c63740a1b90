#!/bin/bash
# the thin-operand weight gradients (wgrad_thinx_kernel: at most 4 input channels; wgrad_thing_kernel: the heads' [pixel][4] gradient carriers) against the kernels they
# replace: correctness on small ragged shapes (CHECK=1: torch's weight gradient of the stored operands) and time at the step's shapes.  HV_WGRAD_THIN: bit 0 / bit 1
for t in 0 3; do
  echo "HV_WGRAD_THIN=$t"
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 2 40 48 4 16 5 1 2 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 3 37 29 4 8 5 1 2 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 3 23 19 128 4 4 1 1 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 2 40 48 12 4 3 1 1 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 3 37 29 8 4 3 1 1 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t python tools/bench_wgrad.py 16 256 256 4 16 5 1 2 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t python tools/bench_wgrad.py 32 31 31 512 4 4 1 1 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t python tools/bench_wgrad.py 16 256 256 12 4 3 1 1 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t python tools/bench_wgrad.py 16 256 256 8 4 3 1 1 2>&1 | grep "^W "
done
