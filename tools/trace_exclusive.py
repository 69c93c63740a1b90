"""Who owns the step's wall time?  From a rocprofv3 --kernel-trace CSV of a graph-replayed bench run: per kernel name, the time during which it
was the ONLY kernel on the GPU (exclusive), the time it shared with others, and its launches -- over the densest window (the timed region).

    python tools/trace_exclusive.py <kernel_trace.csv> [window_ms=150] [steps_in_window]

Exclusive time is what a faster kernel gives back one for one; shared time only matters as far as the GPU was full.
"""
import bisect
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    w = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 150e6
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
    starts = [x[0] for x in iv]
    best, lo = -1, iv[0][0]
    for i in range(0, len(starts), 50):
        n = bisect.bisect_left(starts, starts[i] + w) - i
        if n > best:
            best, lo = n, starts[i]
    sel = [x for x in iv if lo <= x[0] and x[1] < lo + w]
    ev = []
    for i, (s, e, n) in enumerate(sel):
        ev.append((s, 1, i)); ev.append((e, 0, i))
    ev.sort()
    active, last = set(), ev[0][0]
    excl, shared, cnt = collections.Counter(), collections.Counter(), collections.Counter()
    idle = 0
    hist = collections.Counter()
    for t, kind, i in ev:
        dt = t - last
        if dt > 0:
            if not active:
                idle += dt
            elif len(active) == 1:
                excl[sel[next(iter(active))][2]] += dt
            else:
                for j in active:
                    shared[sel[j][2]] += dt / len(active)
            hist[min(len(active), 6)] += dt
        last = t
        if kind:
            active.add(i); cnt[sel[i][2]] += 1
        else:
            active.discard(i)
    span = sel[-1][1] - sel[0][0]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else None
    per = (lambda v: v / steps / 1e3) if steps else (lambda v: v / 1e3)
    unit = 'us/step' if steps else 'us'
    print('window %.1f ms, %d kernels; idle %.1f%%; time with k kernels running: %s' %
          (span / 1e6, len(sel), 100.0 * idle / span, '  '.join('%d: %.1f%%' % (k, 100.0 * v / span) for k, v in sorted(hist.items()))))
    print('%-72s %10s %10s %8s   (%s)' % ('kernel', 'exclusive', 'shared/k', 'launches', unit))
    names = sorted(set(excl) | set(shared), key=lambda n: -(excl[n] + shared[n]))
    for n in names[:45]:
        print('%-72s %10.1f %10.1f %8.1f' % (n, per(excl[n]), per(shared[n]), cnt[n] / steps if steps else cnt[n]))
    print('%-72s %10.1f %10.1f' % ('TOTAL', per(sum(excl.values())), per(sum(shared.values()))))


if __name__ == '__main__':
    main()
