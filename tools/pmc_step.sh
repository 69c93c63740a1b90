#!/bin/bash
# HBM traffic of EVERY kernel of the train step, per kernel instantiation: two rocprofv3 --pmc passes (FETCH_SIZE, then
# WRITE_SIZE - they do not fit one pass; kernel-trace only) over a short single-stream eager bench run, summarised into
# profiles/<tag>_traffic_step.json by tools/pmc_step_summary.py.   usage: tools/pmc_step.sh <tag>   (repo root, GPU box)
tag=${1:-r01}
export TMPDIR=/tmp
out=gpurun_out/traffic_step_$tag
mkdir -p $out
args="bench.py --serial --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-inference --no-extra"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/f -o f --output-format csv -- python3 $args > $out/f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/w -o w --output-format csv -- python3 $args > $out/w.log 2>&1 || exit 1
python3 tools/pmc_step_summary.py $out $out/${tag}_traffic_step.json
