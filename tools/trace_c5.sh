root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/trace_c5; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/ks -o c5 --output-format csv -- python3 $root/bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-dropin > $out/ks.log 2>&1 && echo ok
cd $root
KT=$(find $out/ks -name "c5_kernel_trace.csv" | head -1)
python3 tools/trace_chain.py $KT > $out/c5_chain.txt
python3 tools/trace_leaf.py $KT 20 21 > $out/c5_leaf.txt
cp $(find $out/ks -name "c5_kernel_stats.csv" | head -1) $out/c5_kernel_stats.csv
rm -rf $out/ks
head -30 $out/c5_chain.txt; cat $out/c5_leaf.txt | head -45
