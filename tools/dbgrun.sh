cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 3 7; do
  export MPQR_DBG_SOLVE=$d
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/dbg$d -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --no-cpu-baseline --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/dbg$d.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/dbg$d -name "*kernel_stats.csv" | head -1)
  grep "gh_solve\|t_panel" $f | awk -F, -v d=$d '{print "dbg",d,$1, $(NF-5)}' | cut -c1-160
done
