#!/bin/bash
# in-kernel phase stamps of the fused leaf's kernels (MPQR_KTRACE build on the GPU box).  usage: bash tools/ktrace_fl.sh [m n]
root=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt && mkdir -p /tmp/kt && cp -r $root/mixedprecisionblockqr_amd $root/include /tmp/kt/ && rm -rf /tmp/kt/mixedprecisionblockqr_amd/csrc/build /tmp/kt/mixedprecisionblockqr_amd/libmpqr.so
make -C /tmp/kt/mixedprecisionblockqr_amd/csrc -j16 EXTRA=-DMPQR_KTRACE > /tmp/kt/build.log 2>&1 || { tail -20 /tmp/kt/build.log; exit 1; }
cd /tmp/kt && M=${1:-16384} N=${2:-2048} python3 - <<'PY'
import os
import mixedprecisionblockqr_amd as mp
h = mp.Handle(0)
h.plan(int(os.environ["M"]), int(os.environ["N"]), 128)
h.generate(1234); h.factor(); h.sync()
print(h.timings()["ms_factor"])
h.close()
PY
