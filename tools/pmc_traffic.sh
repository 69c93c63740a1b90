#!/bin/bash
# HBM traffic of one conv / wgrad micro-benchmark from the L2's memory-side counters (MI355X_MICROARCH.md, HBM section):
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes (they do not fit one pass), kernel-trace only.
# usage: tools/pmc_traffic.sh <tag> <bench_conv.py|bench_wgrad.py> args...     (run from the repo root on the GPU box)
tag=$1; shift
prog=$1; shift
export TMPDIR=/tmp
out=gpurun_out/traffic_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/f -o f --output-format csv -- python3 tools/$prog "$@" > $out/f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/w -o w --output-format csv -- python3 tools/$prog "$@" > $out/w.log 2>&1 || exit 1
python3 tools/pmc_summary.py $out
