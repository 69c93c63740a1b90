#!/bin/bash
# Per-kernel averages of the default bench command under rocprofv3 for several settings of ONE environment knob (same box, one call):
#   tools/ab_kernel_stats.sh HV_CA_GRAM_NBY "2 4" "ca_gram_backward|ca_fuse_adj|ca_coef"
knob=$1; vals=$2; pat=$3
export TMPDIR=/tmp
out=gpurun_out/abk_$knob
mkdir -p $out
for v in $vals; do
  export $knob=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_$v -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-extra > $out/p_$v.log 2>&1 || exit 1
  echo "== $knob=$v  $(grep '^{' $out/p_$v.log | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'], 'ms/step under rocprof')")"
  python3 - "$pat" $out/p_$v/p_kernel_stats.csv <<'PY'
import csv, re, sys
pat = re.compile(sys.argv[1])
for r in list(csv.reader(open(sys.argv[2])))[1:]:
    if pat.search(r[0]):
        print('%-90s calls %6s  avg_us %8.2f' % (r[0][:90], r[1], float(r[3]) / 1000))
PY
  rm -rf $out/p_$v
done
