#!/bin/bash
b="python tools/bench_conv.py"
for knob in "HV_S2T=0" "HV_S2T=1"; do
  echo "== $knob"
  env $knob $b 16 64 64 128 64 4 2 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 32 32 256 128 4 2 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 128 128 32 16 3 2 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 128 128 16 16 3 2 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 64 64 64 32 3 2 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 64 64 32 16 3 2 1 1 30 2>/dev/null | tail -1
done
