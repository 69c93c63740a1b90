#!/bin/bash
# the generators' thin 256x256 layers through tools/bench_conv.py (rotating buffers: cold reads, as inside a step), with the epilogues the step uses:
# B H W Cin Cout k s p transposed | environment
run() { env $2 ROTATE=8 python tools/bench_conv.py $1 50 2>&1 | grep -v amdgpu.ids | tail -1; }
for px in 0 255; do
  echo "HV_CONV_PX=$px"
  export HV_CONV_PX=$px
  run "16 256 256 12 1 3 1 1 0" "Y32=1 ACT=clamp BIAS=1"
  run "16 256 256 8 1 3 1 1 0" "Y32=1 ACT=clamp BIAS=1"
  run "16 256 256 4 12 3 1 1 1" "MUL=elu"
  run "16 256 256 4 8 3 1 1 1" "MUL=elu"
  run "16 256 256 16 8 3 1 1 0" "ACT=elu BIAS=1"
  run "16 256 256 8 16 3 1 1 1" "MUL=elu"
  run "16 256 256 8 16 3 1 1 0" "ACT=elu BIAS=1"
  run "16 256 256 16 8 3 1 1 1" "MUL=elu"
  run "16 256 256 4 16 5 1 2 0" "ACT=elu BIAS=1"
done
