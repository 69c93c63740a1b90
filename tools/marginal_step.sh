#!/bin/bash
# What would removing a pass buy at STEP level?  Timing-only runs of the bench with the named passes not launched (HV_DIAG_SKIP: kernels inside C entry
# points; HV_DIAG_SKIP_C: whole C entry points by name -- results are wrong, only the step time means anything); one box, interleaved with the plain run.
# usage: tools/marginal_step.sh [c-entry-point-list ...]   (repo root, GPU box)
run() { HV_DIAG_SKIP=$1 HV_DIAG_SKIP_C=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inference --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %s ms' % ('skip: $1 $2', d['ms_per_step']))"; }
run none
if [ $# -gt 0 ]; then
    for x in "$@"; do run none $x; done
    run none
    exit 0
fi
run norm_apply
run norm_bwd
run wgrad_reduce
run none
run copy_channels
run act_bwd
run norm_apply,norm_bwd,wgrad_reduce,copy_channels,act_bwd
run none
