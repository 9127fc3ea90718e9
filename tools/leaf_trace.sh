#!/bin/bash
# kernel trace of one c4 run -> per-leaf timeline (gpurun_out/<tag>_leaf_timeline.txt) and kernel stats
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/trace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/ks -o c4 --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/ks.log 2>&1 && echo "kernel stats ok"
cd $root
KT=$(find $out/ks -name "c4_kernel_trace.csv" | head -1)
cp $(find $out/ks -name "c4_kernel_stats.csv" | head -1) $out/${tag}_c4_kernel_stats.csv
python3 tools/trace_leaf.py $KT 60 61 63 64 65 > $out/${tag}_c4_leaf_timeline.txt
python3 tools/trace_gaps.py $KT > $out/${tag}_c4_stream_gaps.txt
python3 tools/trace_chain.py $KT > $out/${tag}_c4_chain.txt
gzip -c $KT > $out/${tag}_c4_kernel_trace.csv.gz
rm -rf $out/ks
cat $out/${tag}_c4_leaf_timeline.txt
