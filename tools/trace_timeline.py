"""One step of the graph-replayed bench as a timeline, from a rocprofv3 --kernel-trace CSV: per time bin, how many kernels ran together, how busy
the bin was, and the kernels that owned most of it.  Shows where the step is a chain of small launches (room for independent work beside it).

    python tools/trace_timeline.py <kernel_trace.csv> [bin_us=100] [marker substring=first kernel of the step]

The step is cut between two successive launches of the marker kernel in the densest part of the trace (the timed region).
"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    bin_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 100e3
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
    # the timed region = the densest 60 ms of the trace (graph replays back to back); the marker = a kernel launched once per step there
    import bisect
    starts = [x[0] for x in iv]
    best, at = -1, 0
    for i in range(0, len(starts), 50):
        n = bisect.bisect_left(starts, starts[i] + 60e6) - i
        if n > best:
            best, at = n, i
    tail = iv[at:at + best]
    cnt = collections.Counter(n for _, _, n in tail)
    if len(sys.argv) > 3:
        marker = next(n for n in cnt if sys.argv[3] in n)
    else:
        least = min(v for v in cnt.values() if v >= 4)
        marker = next(n for _, _, n in tail if cnt[n] == least)
    marks = [s for s, _, n in tail if n == marker]
    # a marker launched more than once per step (short gaps inside the step, one long gap around its end): keep the launches that follow a long gap
    gaps = [b - a for a, b in zip(marks, marks[1:])]
    if max(gaps) > 3 * min(gaps):
        marks = [m for m, g in zip(marks[1:], gaps) if g > 0.5 * max(gaps)]
    gaps = sorted(b - a for a, b in zip(marks, marks[1:]))
    gap = gaps[len(gaps) // 2]
    k = next(i for i, (a, b) in enumerate(zip(marks, marks[1:])) if abs((b - a) - gap) < 0.02 * gap)
    lo, hi = marks[k], marks[k + 1]
    sel = [x for x in tail if x[1] > lo and x[0] < hi]
    print('marker %s; step %.3f ms, %d kernels' % (marker, (hi - lo) / 1e6, len(sel)))
    print('%8s %6s %6s  %s' % ('t_us', 'busy%', 'conc', 'owners (share of the bin)'))
    t = lo
    while t < hi:
        t1 = min(t + bin_ns, hi)
        own = collections.Counter()
        edges = []
        for s, e, n in sel:
            a, b = max(s, t), min(e, t1)
            if b > a:
                own[n] += b - a
                edges.append((a, 1)); edges.append((b, -1))
        edges.sort()
        busy, depth, last = 0, 0, t
        for x, d in edges:
            if depth > 0:
                busy += x - last
            last, depth = x, depth + d
        tot = sum(own.values())
        top = '  '.join('%s %.0f%%' % (n, 100.0 * v / (t1 - t)) for n, v in own.most_common(3))
        print('%8.0f %6.0f %6.2f  %s' % ((t - lo) / 1e3, 100.0 * busy / (t1 - t), tot / max(busy, 1), top))
        t = t1


if __name__ == '__main__':
    main()
