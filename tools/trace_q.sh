#!/bin/bash
# every dispatch of the Q-formation phase with the idle time in front of it (kernel trace of the default bench)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktq
rocprofv3 --kernel-trace -d /tmp/ktq -o t --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropin --no-alone > /tmp/ktq.log 2>&1 || { tail -5 /tmp/ktq.log; exit 1; }
python3 $root/tools/trace_qphase.py $(find /tmp/ktq -name '*kernel_trace.csv' | head -1) --all
