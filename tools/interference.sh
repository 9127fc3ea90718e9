#!/bin/bash
# kernel traces of the default bench with 7 / 4 of 8 CUs per shader engine for the far stream -> tools/interference.py
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for rows in 7 4; do
  rm -rf /tmp/kt$rows
  MPQR_UPDATE_CU_ROWS=$rows rocprofv3 --kernel-trace -d /tmp/kt$rows -o t --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropin --no-alone > /tmp/kt$rows.log 2>&1 || { tail -5 /tmp/kt$rows.log; exit 1; }
  f=$(find /tmp/kt$rows -name '*kernel_trace.csv' | head -1)
  echo "=== far stream on $rows of 8 CUs per shader engine" >> $out/interference.txt
  python3 $root/tools/interference.py $f >> $out/interference.txt
done
