#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktc2
rocprofv3 --kernel-trace -d /tmp/ktc2 -o t --output-format csv -- python3 $root/bench.py --config ${1:-c2} --steps 2 --warmup 1 --no-cpu-baseline --no-dropin --no-alone > /tmp/ktc2.log 2>&1 || { tail -5 /tmp/ktc2.log; exit 1; }
python3 $root/tools/trace_all.py $(find /tmp/ktc2 -name '*kernel_trace.csv' | head -1)
