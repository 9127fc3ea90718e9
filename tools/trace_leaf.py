#!/usr/bin/env python3
"""Timeline of single leaves of the panel chain from a rocprofv3 kernel trace: every dispatch (all queues) between
two consecutive gh_solve launches, start offset / duration / queue.  Also the per-leaf period statistics.
usage: trace_leaf.py <kernel_trace.csv> [leaf_index ...]   (indices into the last factorisation's gh_gram launches)"""
import csv, sys, re, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
gen = [i for i, r in enumerate(rows) if 'generate_kernel' in r['Kernel_Name']]
lo = gen[-1]
hi = next((i for i in range(lo, len(rows)) if 'lower_norm' in rows[i]['Kernel_Name'] or 'strip_r' in rows[i]['Kernel_Name']), len(rows))
run = rows[lo + 1:hi]
def short(n):
    n = re.sub(r'^void ', '', n); n = n.replace('mpqr::', '')
    n = re.sub(r'\(.*', '', n)
    return n[:40]
grams = [i for i, r in enumerate(run) if 'gh_solve' in r['Kernel_Name']]      # one solve per leaf
per = [(run[b]['s'] - run[a]['s']) / 1e3 for a, b in zip(grams, grams[1:])]
print("%d leaves; period us: median %.1f mean %.1f min %.1f max %.1f" % (len(grams), statistics.median(per), statistics.mean(per), min(per), max(per)))
print("periods by position in the 8-leaf block (mean us):", [round(statistics.mean(per[k::8]), 1) for k in range(8)])
want = [int(x) for x in sys.argv[2:]] or [len(grams) // 2]
for li in want:
    if li + 1 >= len(grams): continue
    a, b = grams[li], grams[li + 1]
    t0 = run[a]['s']; t1 = run[b]['s']
    print("--- leaf %d: period %.1f us" % (li, (t1 - t0) / 1e3))
    for r in run:
        if r['e'] < t0 or r['s'] > t1: continue
        print("  q%-2s %+8.1f %7.1f  %s" % (r['Queue_Id'], (r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, short(r['Kernel_Name'])))
