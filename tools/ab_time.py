"""tools/ab_time.py -- step time of one configuration on the device-resident API (same-box A/B runs: swap the library or set MPQR_* hooks
between two invocations inside ONE gpurun call).
    python tools/ab_time.py LABEL M N R [REPS]  ->  LABEL  best / median ms_total over REPS steps, ms_factor, ms_panel, ms_form_q of the best"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixedprecisionblockqr_amd as mp

label, m, n, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 7
ob = int(os.environ.get("AB_OUTER_BLOCK", "0"))             # outer block (0: the library's choice)
h = mp.Handle(0)
h.plan(m, n, r, outer_block=ob) if ob else h.plan(m, n, r)
h.generate(1234)
ts = []
for i in range(reps + 2):
    h.factor(); h.sync()
    if i >= 2:
        ts.append(h.timings())
ts.sort(key=lambda t: t["ms_total"])
b, md = ts[0], ts[len(ts) // 2]
mt = h.metrics()
print(f"{label:10s} {m}x{n} r={r}: best {b['ms_total']:.3f} median {md['ms_total']:.3f} ms  (factor {b['ms_factor']:.3f} panel {b['ms_panel']:.3f} q {b['ms_form_q']:.3f})"
      f" far tn {b['ms_far_tn']:.2f} nn {b['ms_far_nn']:.2f} wait {b['ms_chain_wait']:.2f}"
      f"  backward {mt['backward_error']:.2e} orth {mt['q_error_fro']:.2e} gh_leaves {b['n_gh_leaves']}", flush=True)
h.close()
