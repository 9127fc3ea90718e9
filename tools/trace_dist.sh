# kernel trace of the distributed schedule forced onto one GPU (MPQR_FORCE_DIST=1): chain summary + stream gaps
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/trace_dist; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
MPQR_FORCE_DIST=1 rocprofv3 --kernel-trace --stats -d $out/ks -o c4 --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropin --no-alone > $out/ks.log 2>&1 && echo ok
cd $root
KT=$(find $out/ks -name "c4_kernel_trace.csv" | head -1)
python3 tools/trace_chain.py $KT > $out/dist_chain.txt
cp $(find $out/ks -name "c4_kernel_stats.csv" | head -1) $out/dist_kernel_stats.csv
rm -rf $out/ks
head -45 $out/dist_chain.txt
