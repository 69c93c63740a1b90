#!/bin/bash
b="python tools/bench_conv.py"
for knob in "HV_HALO2_RING=4" "HV_HALO2_RING=8" "HV_HALO2_RING=16"; do
  echo "== $knob"
  env $knob $b 16 32 32 256 512 4 1 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 31 31 512 256 4 1 1 1 30 2>/dev/null | tail -1
done
