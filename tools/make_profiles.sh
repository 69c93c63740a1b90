#!/bin/bash
# Refresh profiles/ for a tag (run from the repo root on the GPU box; outputs under gpurun_out/profiles_<tag>/, copy what
# should be judged into profiles/):
#   <tag>_bench_fp16.json / _fp32.json          plain bench lines (with cpu_baseline for fp16)
#   <tag>_bench_fp16_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the default bench command
#   <tag>_serial_kernel_stats.csv               the same with --serial: per-kernel averages without stream overlap
tag=$1
out=gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 3 > $out/${tag}_bench_fp16.log 2>&1 || exit 1
grep '^{' $out/${tag}_bench_fp16.log | tail -1 > $out/${tag}_bench_fp16.json
python3 bench.py --steps 20 --warmup 3 --precision fp32 --no-cpu-baseline --no-inference > $out/${tag}_bench_fp32.log 2>&1 || exit 1
grep '^{' $out/${tag}_bench_fp32.log | tail -1 > $out/${tag}_bench_fp32.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p1 -o p1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-extra > $out/p1.log 2>&1 || exit 1
cp $out/p1/p1_kernel_stats.csv $out/${tag}_bench_fp16_kernel_stats.csv
grep '^{' $out/p1.log | tail -1 > $out/${tag}_bench_fp16_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p2 -o p2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-extra --serial > $out/p2.log 2>&1 || exit 1
cp $out/p2/p2_kernel_stats.csv $out/${tag}_serial_kernel_stats.csv
grep '^{' $out/p2.log | tail -1 > $out/${tag}_bench_fp16_serial_under_rocprof.json
rm -rf $out/p1 $out/p2
ls -la $out
