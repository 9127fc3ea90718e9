#!/bin/bash
# kernel trace + HIP runtime API trace of one c4 run: is the chain stream ever waiting for the HOST (launch issued late)?
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/htrace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-runtime-trace -d $out/ks -o c4 --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropin > $out/ks.log 2>&1 && echo "trace ok"
cd $root
ls -la $out/ks/*/ 2>/dev/null | head; 
for f in $(find $out/ks -name "*.csv"); do gzip -c $f > $out/$(basename $f).gz; done
rm -rf $out/ks
ls -la $out
