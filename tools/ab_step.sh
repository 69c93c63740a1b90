#!/bin/bash
# step-level A/B of one environment switch inside ONE gpurun call (box-to-box variance is larger than most effects):
# usage: tools/ab_step.sh VAR offvalue onvalue [pairs]
var=$1; off=$2; on=$3; pairs=${4:-3}
for i in $(seq $pairs); do
  for v in $off $on; do
    export $var=$v
    printf "%s=%s " $var $v
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inference --no-extra 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])" || exit 1
  done
done
