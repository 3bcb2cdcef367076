"""Overlap states of the pass from a rocprofv3 kernel trace (…_kernel_trace.csv): for how much of the timed span is
nobody tracing, one traversal kernel running, two; what runs beside them.  Usage: python tools/timeline.py trace.csv"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = []
short = lambda n: ('trace' if ('k_wf_trace<false' in n or 'k_wf_trace2<false' in n) else 'shade' if 'k_wf_shade<false' in n else 'gen' if 'k_wf_gen<false' in n
                   else 'finish' if 'k_wf_finish<false' in n else 'resolve' if 'k_wf_resolve' in n else None)
ks = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']) for r in rows]
ks = [k for k in ks if k[0]]
# the timed region: the longest run of launches without a pause of more than 5 ms; its middle 80 %
ks.sort(key=lambda k: k[1])
runs, cur, end = [], [ks[0]], ks[0][2]
for k in ks[1:]:
    if k[1] - end > 5e6: runs.append(cur); cur = []
    cur.append(k); end = max(end, k[2])
runs.append(cur)
ks = max(runs, key=len)
t0 = min(k[1] for k in ks); t1 = max(k[2] for k in ks)
lo, hi = t0 + 0.1 * (t1 - t0), t1 - 0.1 * (t1 - t0)
pts = []
for n, s, e, q in ks:
    s, e = max(s, lo), min(e, hi)
    if e > s: pts += [(s, n, 1), (e, n, -1)]
pts.sort()
state = collections.Counter(); dur = collections.Counter(); last = lo
for t, n, d in pts:
    key = 'trace=%d shade=%d gen=%d other=%d' % (state['trace'], state['shade'], state['gen'], state['finish'] + state['resolve'])
    dur[key] += t - last; last = t
    state[n] += d
tot = sum(dur.values())
for k, v in sorted(dur.items(), key=lambda kv: -kv[1]): print('%-32s %6.2f %%' % (k, 100.0 * v / tot))
per = collections.defaultdict(list)
for n, s, e, q in ks:
    if s >= lo and e <= hi: per[n].append((e - s) / 1e6)
for n, v in per.items(): print('%-8s launches %5d  avg %.3f ms  sum %.1f ms  (span %.1f ms)' % (n, len(v), sum(v) / len(v), sum(v), (hi - lo) / 1e6))
# gaps on each pipe's queue between the end of a kernel and the start of the next one of that queue
gaps = collections.defaultdict(list)
byq = collections.defaultdict(list)
for n, s, e, q in ks:
    if n in ('trace', 'shade', 'gen') and s >= lo and e <= hi: byq[q].append((s, e, n))
for q, v in byq.items():
    v.sort()
    for a, b in zip(v, v[1:]): gaps[q + ':' + a[2] + '->' + b[2]].append((b[0] - a[1]) / 1e3)
for k, v in sorted(gaps.items()): print('gap %-22s n %5d  avg %7.1f us  max %8.1f us' % (k, len(v), sum(v) / len(v), max(v)))
