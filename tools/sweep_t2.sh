#!/bin/bash
# tile / wave variants of the 2x2-tap parity classes (data gradient of the 4x4 stride-2 PatchGAN layers): HV_HALO2_T2
b="python tools/bench_conv.py"
for v in ${VARIANTS:-0 1}; do
  echo "== HV_HALO2_T2=$v"
  env HV_HALO2_T2=$v $b 16 64 64 128 64 4 2 1 1 30 2>/dev/null | tail -1
  env HV_HALO2_T2=$v $b 16 32 32 256 128 4 2 1 1 30 2>/dev/null | tail -1
  env HV_HALO2_T2=$v $b 16 16 16 512 256 4 2 1 1 30 2>/dev/null | tail -1
done
