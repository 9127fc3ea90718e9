# usage: bash tools/prof_trace.sh <tag> [bench args...] -- kernel trace of a short bench run -> gpurun_out/<tag>_trace.csv (+ stats)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o t --output-format csv -- python3 $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
cp $(find gpurun_out/prof_$tag -name "t_kernel_trace.csv" | head -1) gpurun_out/${tag}_trace.csv
cp $(find gpurun_out/prof_$tag -name "t_kernel_stats.csv" | head -1) gpurun_out/${tag}_stats.csv
rm -rf gpurun_out/prof_$tag
python3 tools/trace_gaps.py gpurun_out/${tag}_trace.csv > gpurun_out/${tag}_gaps.txt 2>&1
tail -40 gpurun_out/${tag}_gaps.txt
