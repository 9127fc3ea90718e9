#!/usr/bin/env python3
"""Critical-stream analysis of a rocprofv3 kernel trace: busy time and idle gaps per kernel on each queue.
usage: trace_gaps.py <kernel_trace.csv> [n_runs]   (the trace holds n_runs factorizations; the last one is analysed)"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# split into factorizations at generate_kernel launches
gen = [i for i, r in enumerate(rows) if 'generate_kernel' in r['Kernel_Name']]
lo = gen[-1]
hi = next((i for i in range(lo, len(rows)) if 'lower_norm' in rows[i]['Kernel_Name'] or 'strip_r' in rows[i]['Kernel_Name']), len(rows))
run = rows[lo + 1:hi]
def short(n):
    n = re.sub(r'^void ', '', n); n = n.replace('mpqr::', '')
    return n[:44]
t0, t1 = run[0]['s'], max(r['e'] for r in run)
print("window %.2f ms, %d dispatches" % ((t1 - t0) / 1e6, len(run)))
byq = collections.defaultdict(list)
for r in run: byq[r['Queue_Id']].append(r)
for q, lst in byq.items():
    busy = sum(r['e'] - r['s'] for r in lst)
    print("queue %s: %d kernels, busy %.2f ms" % (q, len(lst), busy / 1e6))
# union busy over all queues -> idle time of the whole GPU
ev = sorted((r['s'], r['e']) for r in run)
cur_s, cur_e = ev[0]; tot = 0
for s, e in ev[1:]:
    if s > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
tot += cur_e - cur_s
print("GPU busy (any queue) %.2f ms, idle %.2f ms" % (tot / 1e6, (t1 - t0 - tot) / 1e6))
# main queue = the one with most kernels: gap before each kernel attributed to that kernel's name
mq = max(byq, key=lambda q: len(byq[q])); lst = byq[mq]
gap = collections.Counter(); cnt = collections.Counter(); dur = collections.Counter()
for a, b in zip(lst, lst[1:]):
    g = b['s'] - a['e']
    gap[short(b['Kernel_Name'])] += max(g, 0)
for r in lst: cnt[short(r['Kernel_Name'])] += 1; dur[short(r['Kernel_Name'])] += r['e'] - r['s']
print("main queue %s: kernel / calls / busy ms / gap-before ms / avg gap us" % mq)
for k, v in sorted(dur.items(), key=lambda kv: -(kv[1] + gap[kv[0]])):
    print("  %-44s %5d %8.2f %8.2f %7.1f" % (k, cnt[k], v / 1e6, gap[k] / 1e6, gap[k] / 1e3 / max(cnt[k], 1)))
print("  total busy %.2f gap %.2f" % (sum(dur.values()) / 1e6, sum(gap.values()) / 1e6))
