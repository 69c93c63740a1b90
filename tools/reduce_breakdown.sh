#!/bin/bash
# per-dispatch durations of one kernel family grouped by the kernel launched before it (which layer a slab reduction belongs to)
# usage: tools/reduce_breakdown.sh [kernel-name-substring]     (run on the GPU box from the repo root)
pat=${1:-wgrad_reduce_kernel}
export TMPDIR=/tmp
out=gpurun_out/rb
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-inference --no-extra --serial > $out/log.txt 2>&1 || exit 1
python3 - "$pat" $out/t_kernel_trace.csv <<'PY'
import csv, sys, collections
pat, path = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.OrderedDict()
prev = None
for r in rows:
    if pat in r['Kernel_Name']:
        key = (prev['Kernel_Name'][:70] if prev else '-', prev['Grid_Size_X'] if prev else 0, r['Grid_Size_X'])
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if prev: a[2] += (int(prev['End_Timestamp']) - int(prev['Start_Timestamp'])) / 1e3
    prev = r
tot = sum(a[1] for a in agg.values())
print('total %.1f us over %d dispatches' % (tot, sum(a[0] for a in agg.values())))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-72s prevgrid %8s grid %8s  n %4d  avg %7.1f us (prev %7.1f us)  share %5.1f%%' % (k[0], k[1], k[2], a[0], a[1] / a[0], a[2] / a[0], 100 * a[1] / tot))
PY
