cd /tmp; export TMPDIR=/tmp
for shp in "1 16 16 64 64 3 1 1" "16 16 16 64 64 3 1 1" "16 32 32 64 64 3 1 1" "16 64 64 64 64 3 1 1" "16 128 128 64 64 3 1 1"; do
  tag=$(echo $shp | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/wgfix_$tag -o p --output-format csv -- python3 /root/repo/tools/bench_wgrad.py $shp 30 > /dev/null 2>&1
  python3 - <<PY
import csv
out=[]
for r in csv.DictReader(open('/root/repo/gpurun_out/wgfix_$tag/p_kernel_stats.csv')):
    if 'wgrad' in r['Name']: out.append('%s %.2f us' % (r['Name'][5:34], float(r['AverageNs'])/1e3))
print('$shp:', '  '.join(out))
PY
done
