"""One graph-replayed step as the list of its kernels in start order, from a rocprofv3 --kernel-trace CSV: start (us from the step's first kernel), duration,
queue, name -- plus, per queue, busy time and the idle gaps between its kernels.   python tools/step_sequence.py <kernel_trace.csv> [marker=gen_input_kernel]"""
import collections, csv, re, sys, bisect
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'gen_input_kernel'
def short(n):
    n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*$', '', n); return n[:52]
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')) for r in rows)
starts = [x[0] for x in iv]
best, at = -1, 0
for i in range(0, len(starts), 50):
    n = bisect.bisect_left(starts, starts[i] + 60e6) - i
    if n > best: best, at = n, i
tail = iv[at:at + best]
marks = [s for s, _, n, _ in tail if marker in n]
# gen_input runs twice per step (coarse, fine): keep every other one
gaps = [b - a for a, b in zip(marks, marks[1:])]
if gaps and max(gaps) > 3 * min(gaps): marks = [m for m, g in zip(marks[1:], gaps) if g > 0.5 * max(gaps)]
k = len(marks) // 2
lo, hi = marks[k], marks[k + 1]
sel = [x for x in tail if lo <= x[0] < hi]
print('step %.3f ms, %d kernels' % ((hi - lo) / 1e6, len(sel)))
qs = collections.OrderedDict()
for s, e, n, q in sel: qs.setdefault(q, []).append((s, e, n))
for q, lst in qs.items():
    busy = sum(e - s for s, e, _ in lst)
    print('queue %s: %d kernels, busy %.1f us' % (q, len(lst), busy / 1e3))
end_prev = {}
for s, e, n, q in sel:
    gap = (s - end_prev[q]) / 1e3 if q in end_prev else 0.0
    end_prev[q] = e
    print('%9.1f %8.1f  q%-3s gap %7.1f  %s' % ((s - lo) / 1e3, (e - s) / 1e3, q, gap, n))
