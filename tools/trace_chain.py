#!/usr/bin/env python3
"""Where the panel chain's time goes, from a rocprofv3 kernel trace of bench.py: the chain queue of the LAST factorisation,
busy time per kernel, idle time charged to the kernel that follows it, split by the leaf's position in its 8-leaf block, and
the longest idle stretches with what the other queues ran meanwhile.
usage: trace_chain.py <kernel_trace.csv>"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ab = [i for i, r in enumerate(rows) if 'absmax' in r['Kernel_Name']]          # compute_scale opens every mpqr_factor
run = rows[ab[-1]:]
def short(n):
    n = re.sub(r'^void ', '', n); n = n.replace('mpqr::', ''); n = re.sub(r'^_ZN4mpqr\d+', '', n)
    n = re.sub(r'\(.*', '', n); n = re.sub(r'EP[KfD].*', '', n)
    return n[:28]
solves = [r for r in run if 'gh_solve' in r['Kernel_Name']]
cq = solves[0]['Queue_Id']
chain = [r for r in run if r['Queue_Id'] == cq]
first = next(i for i, r in enumerate(chain) if 'gh_gram' in r['Kernel_Name'])
last = max(i for i, r in enumerate(chain) if 'gh_apply' in r['Kernel_Name'] or 'leaf_a_kernel' in r['Kernel_Name'])
chain = chain[first:last + 1]
span = (chain[-1]['e'] - chain[0]['s']) / 1e3
solves = [r for r in chain if 'gh_solve' in r['Kernel_Name']]      # (bench.py's stand-alone solve timing comes after the slice)
print(f"chain queue {cq}: {len(chain)} dispatches, {len(solves)} leaves, span {span / 1e3:.2f} ms = {span / len(solves):.1f} us per leaf")
busy = collections.defaultdict(float); gap = collections.defaultdict(float); cnt = collections.defaultdict(int)
pos_gap = collections.defaultdict(float); pos_busy = collections.defaultdict(float)
leaf = -1; gaps = []
prev_e = chain[0]['s']
for r in chain:
    n = short(r['Kernel_Name'])
    if 'gh_reduce_kernel' in r['Kernel_Name']: leaf += 1      # (one per leaf: behind gh_gram, or behind the fused leaf's leaf_b)
    g = max(0, r['s'] - prev_e) / 1e3; d = (r['e'] - r['s']) / 1e3
    busy[n] += d; gap[n] += g; cnt[n] += 1
    pos_gap[leaf % 8] += g; pos_busy[leaf % 8] += d
    if g > 25: gaps.append((g, leaf, n, r['s'] - int(g * 1e3), r['s']))
    prev_e = max(prev_e, r['e'])
print("%-30s %6s %9s %9s %9s" % ("kernel", "calls", "busy ms", "idle ms", "avg idle us"))
for n in sorted(busy, key=lambda k: -(busy[k] + gap[k])):
    print("%-30s %6d %9.2f %9.2f %9.1f" % (n, cnt[n], busy[n] / 1e3, gap[n] / 1e3, gap[n] / cnt[n]))
print("total busy %.2f ms, idle %.2f ms" % (sum(busy.values()) / 1e3, sum(gap.values()) / 1e3))
grams = [r for r in chain if 'gh_solve' in r['Kernel_Name']]
per = [(b['s'] - a['s']) / 1e3 for a, b in zip(grams, grams[1:])]
print("leaf periods (gh_solve to gh_solve, us) per top-level block of 8 leaves:")
for blk in range(0, (len(per) + 7) // 8):
    p8 = per[8 * blk:8 * blk + 8]
    print("  block %2d: %s  sum %.0f" % (blk, " ".join("%4.0f" % x for x in p8), sum(p8)))
print("idle stretches > 25 us: %d, total %.2f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e3))
for g, lf, n, a, b in sorted(gaps, reverse=True)[:6]:
    print(f"--- {g:.0f} us idle before {n} of leaf {lf} (position {lf % 8}); other queues meanwhile:")
    for r in run:
        if r['Queue_Id'] != cq and r['e'] > a and r['s'] < b:
            print("     q%-2s %+8.1f %7.1f  %s" % (r['Queue_Id'], (r['s'] - a) / 1e3, (r['e'] - r['s']) / 1e3, short(r['Kernel_Name'])))
