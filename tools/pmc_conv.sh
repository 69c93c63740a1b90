#!/bin/bash
# PMC passes over one conv / wgrad micro-benchmark.  usage: tools/pmc_conv.sh <tag> <bench_conv.py|bench_wgrad.py> args...
# (run from the repo root on the GPU box; counters in their own rocprofv3 runs, no trace domains besides kernel-trace)
tag=$1; shift
prog=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $out/p1 -o p1 --output-format csv -- python3 tools/$prog "$@" > $out/p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $out/p2 -o p2 --output-format csv -- python3 tools/$prog "$@" > $out/p2.log 2>&1 || exit 1
python3 tools/pmc_summary.py $out
