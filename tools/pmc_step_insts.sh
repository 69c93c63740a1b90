#!/bin/bash
# Instruction counts of EVERY kernel of the train step (one rocprofv3 --pmc pass over a short single-stream eager bench run: vector / scalar / MFMA / LDS /
# vector-memory instructions and waves per launch), next to the launch's duration under the counters: which kernels are bound by instruction issue rather than
# bytes.  usage: tools/pmc_step_insts.sh <tag>   (repo root, GPU box) -> gpurun_out/insts_step_<tag>.txt
tag=${1:-r00}
export TMPDIR=/tmp
out=gpurun_out/insts_step_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d $out/p -o p --output-format csv -- python3 bench.py --serial --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-inference --no-extra > $out/p.log 2>&1 || exit 1
python3 - $out/p/p_counter_collection.csv > gpurun_out/insts_step_$tag.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
disp = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[(k, r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
n = defaultdict(int); dur = defaultdict(float)
for (k, d), us in disp.items():
    n[k] += 1; dur[k] += us
rows = []
for k in acc:
    c = acc[k]; m = n[k]
    valu, salu = c['SQ_INSTS_VALU'] / m, c['SQ_INSTS_SALU'] / m
    # a SIMD issues one vector instruction of a wave every 4 cycles; 1 024 SIMDs; 2.1 GHz under load
    issue_us = (valu * 4 + c['SQ_INSTS_MFMA'] / m * 16) / 1024 / 2100.0
    rows.append((dur[k], k, m, dur[k] / m, valu, salu, c['SQ_INSTS_MFMA'] / m, c['SQ_INSTS_LDS'] / m, c['SQ_INSTS_VMEM_RD'] / m, c['SQ_WAVES'] / m, issue_us))
print('%-64s %5s %8s %9s %9s %8s %8s %8s %7s %8s' % ('kernel', 'n', 'us', 'VALU', 'SALU', 'MFMA', 'LDS', 'VMEM_RD', 'waves', 'issue_us'))
for t, k, m, us, valu, salu, mf, lds, vm, wv, iss in sorted(rows, reverse=True):
    print('%-64s %5d %8.1f %9.0f %9.0f %8.0f %8.0f %8.0f %7.0f %8.1f' % (k[:64], m, us, valu, salu, mf, lds, vm, wv, iss))
PY
head -60 gpurun_out/insts_step_$tag.txt
