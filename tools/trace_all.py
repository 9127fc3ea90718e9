#!/usr/bin/env python3
"""Every dispatch of the last factorisation in a rocprofv3 kernel trace, all queues, with start offset, duration and the idle time of ITS
queue in front of it.  usage: trace_all.py <kernel_trace.csv> [max_rows]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ab = [i for i, r in enumerate(rows) if 'absmax' in r['Kernel_Name']]
bounds = ab + [len(rows)]
run = None
for a, b in reversed(list(zip(bounds, bounds[1:]))):
    if sum(1 for r in rows[a:b] if 'gh_solve' in r['Kernel_Name']) >= 8:
        hi = next((i for i in range(a, b) if 'lower_norm' in rows[i]['Kernel_Name'] or 'strip_r' in rows[i]['Kernel_Name']), b)
        run = rows[a:hi]; break
t0 = run[0]['s']; last = {}
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
for r in run[:n]:
    q = r['Queue_Id']; gap = (r['s'] - last[q]) / 1e3 if q in last else 0.0
    last[q] = max(last.get(q, 0), r['e'])
    name = re.sub(r'^void ', '', r['Kernel_Name']).replace('mpqr::', ''); name = re.sub(r'\(.*', '', name)[:44]
    print("%9.1f us %8.1f us  gap %7.1f  q%s %s" % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, gap, q, name))
print("total %.1f us" % ((max(r['e'] for r in run) - t0) / 1e3))
