"""GPU idle analysis of a rocprofv3 --kernel-trace CSV: busy union, concurrency, largest idle gaps by (prev -> next) kernel.

    python tools/trace_gaps.py <kernel_trace.csv> [tail_fraction=0.25 | dense:<ms>]

dense:<ms> analyses the <ms>-long window that holds the most kernel launches -- in a bench.py trace that is the graph-replayed timed
region (the eager survey / roofline steps around it launch far fewer kernels per millisecond).
"""
import bisect
import collections
import csv
import statistics
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    arg = sys.argv[2] if len(sys.argv) > 2 else '0.25'
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in rows)
    t0, t1 = iv[0][0], max(e for _, e, _ in iv)
    if arg.startswith('dense:'):
        w = float(arg[6:]) * 1e6
        starts = [x[0] for x in iv]
        best, lo = -1, t0
        for i, s0 in enumerate(starts):
            n = bisect.bisect_left(starts, s0 + w) - i
            if n > best:
                best, lo = n, s0
        sel = [x for x in iv if lo <= x[0] < lo + w]
    else:
        lo = t0 + (t1 - t0) * (1 - float(arg))
        sel = [x for x in iv if x[0] >= lo]
    span = max(e for _, e, _ in sel) - sel[0][0]
    busy, tot = 0, sum(e - s for s, e, _ in sel)
    cs, ce = sel[0][0], sel[0][1]
    gaps = []
    agg = collections.defaultdict(lambda: [0, 0])
    prev = sel[0][2]
    for s, e, n in sel[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append(s - ce)
            if s - ce > 10000:
                agg[(prev, n)][0] += 1
                agg[(prev, n)][1] += s - ce
            cs, ce, prev = s, e, n
        elif e > ce:
            ce, prev = e, n
    busy += ce - cs
    print('kernels %d  span %.2f ms  busy(union) %.2f ms  sum %.2f ms  idle %.1f%%  avg concurrency %.2f' %
          (len(sel), span / 1e6, busy / 1e6, tot / 1e6, 100 * (1 - busy / span), tot / busy))
    print('gaps: n %d total %.2f ms median %.1f us; >10us: n %d total %.2f ms' %
          (len(gaps), sum(gaps) / 1e6, statistics.median(gaps) / 1e3, sum(1 for g in gaps if g > 10000), sum(g for g in gaps if g > 10000) / 1e6))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
        print('%8.1f us  n=%3d  %s -> %s' % (v[1] / 1e3, v[0], k[0], k[1]))


if __name__ == '__main__':
    main()
