"""Per-step summary of a rocprofv3 --stats kernel_stats.csv: python tools/stats_summary.py file.csv steps [top]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6
n = sum(int(r['Calls']) for r in rows) / steps
print('kernel time per step %.3f ms, %.0f launches' % (tot, n))
fam = collections.defaultdict(lambda: [0, 0])
for r in rows:
    m = re.match(r'(?:void )?(?:_Z\d+)?([A-Za-z_0-9]+?)(?:_kernel)?(?:I|<|\()', r['Name'])
    k = m.group(1) if m else r['Name'][:30]
    fam[k][0] += float(r['TotalDurationNs']) / steps / 1e6; fam[k][1] += int(r['Calls']) / steps
print('--- families')
for k, v in sorted(fam.items(), key=lambda x: -x[1][0])[:28]:
    print('%-32s %7.3f ms %6.1f launches' % (k, v[0], v[1]))
print('--- instantiations')
for r in rows[:top]:
    print('%-100s %6.1f /step %7.3f ms  %7.1f us' % (r['Name'][:100], int(r['Calls']) / steps, float(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e3))
