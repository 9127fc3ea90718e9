#!/bin/bash
# in-kernel phase stamps of gh_solve3 / tri_inverse / gh_gram / gh_apply / t_panel (MPQR_KTRACE build, run on the GPU box):
# builds a tracing copy of the library under /tmp and runs one 16384 x 2048 factorisation with it.  usage: bash tools/ktrace_solve.sh
root=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt && mkdir -p /tmp/kt && cp -r $root/mixedprecisionblockqr_amd $root/include /tmp/kt/ && rm -rf /tmp/kt/mixedprecisionblockqr_amd/csrc/build /tmp/kt/mixedprecisionblockqr_amd/libmpqr.so
make -C /tmp/kt/mixedprecisionblockqr_amd/csrc -j16 EXTRA=-DMPQR_KTRACE > /tmp/kt/build.log 2>&1 || { tail -20 /tmp/kt/build.log; exit 1; }
cd /tmp/kt && python3 - <<'PY'
import mixedprecisionblockqr_amd as mp
h = mp.Handle(0)
h.plan(16384, 2048, 128)
h.generate(1234); h.factor(); h.sync()
print(h.timings()["ms_factor"])
h.close()
PY
