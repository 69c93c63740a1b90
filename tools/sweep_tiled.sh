#!/bin/bash
# A/B of the MFMA-fragment-ordered filter tables (hv_conv_desc.w_f16_tiled) on the layers that fetch filters straight into MFMA registers
b="python tools/bench_conv.py"
for knob in "HV_W_TILED=0" "HV_W_TILED=1"; do
  echo "== $knob"
  env $knob $b 16 32 32 256 512 4 1 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 32 32 512 256 4 1 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 64 64 128 256 4 2 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 64 64 64 128 4 1 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 64 64 64 64 3 1 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 256 256 16 16 3 1 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 32 32 256 128 4 2 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 64 64 64 32 3 2 1 1 30 2>/dev/null | tail -1
done
