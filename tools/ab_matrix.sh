#!/bin/bash
# step time over a matrix of two environment switches inside ONE gpurun call: tools/ab_matrix.sh VAR1 "v1 v2 .." VAR2 "w1 w2 .." [rounds]
v1=$1; l1=$2; v2=$3; l2=$4; rounds=${5:-2}
for r in $(seq $rounds); do
  for a in $l1; do for b in $l2; do
    export $v1=$a $v2=$b
    printf "%s=%s %s=%s " $v1 $a $v2 $b
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inference --no-extra 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])" || exit 1
  done; done
done
