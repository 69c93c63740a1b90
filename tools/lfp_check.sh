#!/bin/bash
# conv_lfp_kernel (resident form) against conv_lf_kernel / conv_halo2_kernel on multi-round layer shapes: bit equality and time, both settings of HV_LF_PERSIST
for shape in "16 128 128 64 64" "16 128 128 64 32" "16 128 128 64 16" "16 120 136 64 64" "5 100 100 64 64" "16 128 128 32 32"; do
  for pz in 0 1; do
    echo "== $shape HV_LF_PERSIST=$pz"
    HV_LF_PERSIST=$pz python tools/lf_check.py $shape || exit 1
  done
done
