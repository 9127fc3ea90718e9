#!/usr/bin/env python3
"""Q-formation phase of the last factorisation in a rocprofv3 kernel trace: per-kernel time after the last panel solve.
usage: trace_qphase.py <kernel_trace.csv>"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ab = [i for i, r in enumerate(rows) if 'absmax' in r['Kernel_Name']]          # compute_scale opens every mpqr_factor
# the last run that is a whole factorisation (bench.py's stand-alone timings behind the steps also start with an absmax pass)
bounds = ab + [len(rows)]
run = None
for a, b in reversed(list(zip(bounds, bounds[1:]))):
    if sum(1 for r in rows[a:b] if 'gh_apply' in r['Kernel_Name'] or 'leaf_a_kernel' in r['Kernel_Name']) >= 8:
        hi = next((i for i in range(a, b) if 'lower_norm' in rows[i]['Kernel_Name'] or 'strip_r' in rows[i]['Kernel_Name']), b)
        run = rows[a:hi]
        break
if run is None:
    sys.exit("no factorisation found in the trace")
# Q formation = everything behind the last gh_apply of the run (round 4: Q = I is set up at the START of the factorisation, the identity
# kernels no longer mark the phase)
last_apply = max(i for i, r in enumerate(run) if 'gh_apply' in r['Kernel_Name'] or 'leaf_a_kernel' in r['Kernel_Name'] or 'leaf_tail' in r['Kernel_Name'])
q = run[last_apply + 1:]
first = next((i for i, r in enumerate(q) if 'gemm6_f16_kernel<1, 0' in r['Kernel_Name'] or 'gemm2_f16_kernel' in r['Kernel_Name']), 0)   # first X = Q2^T V
q = q[max(first - 2, 0):]
cut = next((i for i, r in enumerate(q) if 'gh_solve' in r['Kernel_Name'] or 'absmax' in r['Kernel_Name']), len(q))   # (bench.py's stand-alone solve timing follows)
q = q[:cut]
lastg = max((i for i, r in enumerate(q) if 'gemm' in r['Kernel_Name']), default=len(q) - 1)      # (memsets of the metric pass may follow)
q = q[:lastg + 1]
t0, t1 = q[0]['s'], max(r['e'] for r in q)
print("Q phase %.2f ms, %d dispatches" % ((t1 - t0) / 1e6, len(q)))
d = collections.Counter(); c = collections.Counter()
for r in q:
    n = re.sub(r'^void ', '', r['Kernel_Name']).replace('mpqr::', '')[:60]
    d[n] += r['e'] - r['s']; c[n] += 1
for k, v in d.most_common():
    print("  %-60s %4d %8.3f ms" % (k, c[k], v / 1e6))
show_all = len(sys.argv) > 2 and sys.argv[2] == '--all'      # every dispatch with the idle time in front of it
prev = t0
for r in q:
    if show_all:
        print("   %+9.1f us %8.1f us  gap %6.1f  q%s %s" % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, (r['s'] - prev) / 1e3, r['Queue_Id'], re.sub(r'^void ', '', r['Kernel_Name']).replace('mpqr::', '')[:70]))
        prev = max(prev, r['e'])
        continue
    if (r['e'] - r['s']) > 2e5:
        print("   %+9.1f us %8.1f us  %s" % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, re.sub(r'^void ', '', r['Kernel_Name'])[:50]))
