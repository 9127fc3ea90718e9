root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/trace_c2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/ks -o c2 --output-format csv -- python3 $root/bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-dropin > $out/ks.log 2>&1 && echo ok
cd $root
KT=$(find $out/ks -name "c2_kernel_trace.csv" | head -1)
python3 tools/trace_chain.py $KT > $out/c2_chain.txt
python3 tools/trace_leaf.py $KT 3 4 8 > $out/c2_leaf.txt
gzip -c $KT > $out/c2_kernel_trace.csv.gz; rm -rf $out/ks
head -40 $out/c2_chain.txt
