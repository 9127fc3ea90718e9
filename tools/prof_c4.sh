# usage: bash tools/prof_c4.sh [config] -- per-kernel time (ms per factorization, us per call) under rocprofv3
cfg=${1:-c4}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof -o $cfg --output-format csv -- python3 $root/bench.py --config $cfg --no-cpu-baseline --steps 3 --warmup 1 > $root/gpurun_out/prof_$cfg.log 2>&1
cd $root
python3 - "$(find gpurun_out/prof -name "${cfg}_kernel_stats.csv" | head -1)" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]: print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), "%9.2f"%(float(r["TotalDurationNs"])/4e6), "%8.1f"%(float(r["AverageNs"])/1e3))
PY
