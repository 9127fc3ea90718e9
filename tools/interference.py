#!/usr/bin/env python3
"""What the far-update GEMMs cost the panel chain's kernels, from a rocprofv3 kernel trace of bench.py (last factorisation):
every launch of the chain's and the T stream's kernels is classed by how much of it ran while a 256-tile GEMM (gemm6) of another queue
was running, and the durations of the two classes are compared kernel by kernel.
usage: interference.py <kernel_trace.csv>"""
import csv, sys, re, collections, bisect
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ab = [i for i, r in enumerate(rows) if 'absmax' in r['Kernel_Name']]
run = rows[ab[-1]:]
def short(n):
    n = re.sub(r'^void ', '', n); n = n.replace('mpqr::', ''); n = re.sub(r'^_ZN4mpqr\d+', '', n)
    n = re.sub(r'\(.*', '', n); n = re.sub(r'EP[KfD].*', '', n); n = re.sub(r'ENS_.*', '', n)
    return n[:30]
cq = next(r['Queue_Id'] for r in run if 'gh_solve' in r['Kernel_Name'])
last_leaf = max(r['s'] for r in run if 'leaf_a_kernel' in r['Kernel_Name'] or 'gh_apply' in r['Kernel_Name'])
big = [(r['s'], r['e']) for r in run if 'gemm6' in r['Kernel_Name'] and r['Queue_Id'] != cq and r['e'] - r['s'] > 150e3 and r['s'] < last_leaf]
big.sort()
def overlap(s, e):
    t = 0
    for a, b in big:
        if b <= s: continue
        if a >= e: break
        t += min(e, b) - max(s, a)
    return t / max(e - s, 1)
stat = collections.defaultdict(lambda: [[], []])
for r in run:
    if r['s'] > last_leaf or 'gemm6' in r['Kernel_Name']: continue
    f = overlap(r['s'], r['e'])
    q = 'chain' if r['Queue_Id'] == cq else 'other'
    if f >= 0.9: stat[(q, short(r['Kernel_Name']))][1].append((r['e'] - r['s']) / 1e3)
    elif f == 0: stat[(q, short(r['Kernel_Name']))][0].append((r['e'] - r['s']) / 1e3)
print("big GEMMs beside the chain: %d launches, %.2f ms" % (len(big), sum(b - a for a, b in big) / 1e6))
print("%-6s %-30s %14s %14s %7s" % ("queue", "kernel", "clear: n  us", "beside: n  us", "ratio"))
med = lambda v: sorted(v)[len(v) // 2]
for k in sorted(stat, key=lambda k: (k[0], -len(stat[k][0]) - len(stat[k][1]))):
    c, b = stat[k]
    if len(c) < 3 or len(b) < 3: continue
    print("%-6s %-30s %5d %8.1f %5d %8.1f %7.2f" % (k[0], k[1], len(c), med(c), len(b), med(b), med(b) / med(c)))
