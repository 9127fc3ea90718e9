#!/bin/bash
# Collects everything profiles/ holds for one round, on the GPU box:  bash tools/collect_profiles.sh <tag>
# (kernel stats, stream gaps, HBM counters in two separate PMC passes, SQ stall counters, L2 hit rates, bench lines).
# Outputs: gpurun_out/final/ ; copy what is to be judged into profiles/.
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# (--pmc passes serialise kernels: a polling wait can never be satisfied there, so they run the event-ordered schedule, MPQR_TPOLL=0 -- the same GEMM launches)
rocprofv3 --kernel-trace --stats -d $out/ks -o c4 --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-alone > $out/ks.log 2>&1 && echo "kernel stats ok"
MPQR_TPOLL=0 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o c4 --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-alone --dump-records $out/records.json > $out/fetch.log 2>&1 && echo "fetch ok"
MPQR_TPOLL=0 rocprofv3 --pmc WRITE_SIZE -d $out/write -o c4 --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-alone > $out/write.log 2>&1 && echo "write ok"
cd $root
bash tools/pmc_sq.sh c4 > $out/${tag}_c4_sq_stalls.txt 2>&1 && echo "sq ok"
bash tools/pmc_l2.sh $tag > $out/${tag}_c4_l2_hit.txt 2>&1 && echo "l2 ok"
cp $(find $out/ks -name "c4_kernel_stats.csv" | head -1) $out/${tag}_c4_kernel_stats.csv
KT=$(find $out/ks -name "c4_kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $KT > $out/${tag}_c4_stream_gaps.txt
python3 tools/trace_leaf.py $KT 100 101 63 64 65 > $out/${tag}_c4_leaf_timeline.txt      # steady-state panels and a block boundary
python3 tools/trace_chain.py $KT > $out/${tag}_c4_chain.txt                          # chain queue: busy / idle per kernel, leaf periods per block
python3 tools/trace_qphase.py $KT > $out/${tag}_c4_q_phase.txt
F=$(find $out/fetch -name "c4_counter_collection.csv" | head -1); W=$(find $out/write -name "c4_counter_collection.csv" | head -1)
grep -E "Kernel_Name|gemm6_f16|gemm2_f16|gemm_f16_kernel<0, 2>" $F > $out/${tag}_c4_pmc_FETCH_SIZE_gemm.csv
grep -E "Kernel_Name|gemm6_f16|gemm2_f16|gemm_f16_kernel<0, 2>" $W > $out/${tag}_c4_pmc_WRITE_SIZE_gemm.csv
# measured vs algorithmic bytes of the read-modify-write GEMM launches, matched launch by launch against the library's records (--dump-records above)
python3 tools/pmc_traffic.py $F $W $out/records.json $out/${tag}_traffic_rmw.json > $out/${tag}_traffic_rmw_summary.txt
cp $out/${tag}_traffic_rmw.json profiles/${tag}_traffic_rmw.json              # bench.py reads roofline.traffic from here
python3 bench.py > $out/${tag}_c4_bench.json 2> $out/bench_c4.err && echo "bench c4 ok"
python3 bench.py --alone --no-cpu-baseline --no-dropin > $out/${tag}_c4_bench_alone.json 2>> $out/bench_c4.err && echo "bench c4 + kernel-alone ok"
python3 bench.py --config c2 --no-cpu-baseline > $out/${tag}_c2_bench.json 2> $out/bench_c2.err && echo "bench c2 ok"
python3 bench.py --config c3 --no-cpu-baseline > $out/${tag}_c3_bench.json 2> $out/bench_c3.err && echo "bench c3 ok"
python3 bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_c5_bench.json 2> $out/bench_c5.err && echo "bench c5 (fp8) ok"
python3 bench.py --config c5 --precision fp16 --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_c5_fp16_bench.json 2> $out/bench_c5b.err && echo "bench c5 (fp16) ok"
MPQR_FORCE_DIST=1 python3 bench.py --no-cpu-baseline > $out/${tag}_c4_forced_dist_n1_bench.json 2> $out/bench_dist.err && echo "bench forced dist ok"
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "rank_deficient or flagged" 2>&1 | grep -E "ratio|passed|failed" > $out/${tag}_restart_cost.txt && echo "restart cost ok"
MPQR_DBG_BLOCKS=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-alone 2>&1 | grep "mpqr:" > $out/${tag}_c4_blocks_unprofiled.txt && echo "blocks ok"
MPQR_DBG_STAMPS=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-alone 2>&1 | grep "mpqr:" > $out/${tag}_c4_leaf_stamps.txt && echo "stamps ok"
python3 tools/bench_gemm.py > $out/${tag}_gemm_alone.txt 2>&1 && echo "gemm alone ok"
[ -x tools/ubench_mfma.bin ] && ./tools/ubench_mfma.bin > $out/${tag}_mfma_ubench.txt 2>&1 && echo "mfma ubench ok"
python3 tools/precision_study.py $out/${tag}_precision_study.md > /dev/null 2>&1 && echo "precision study ok"
rm -rf $out/ks $out/fetch $out/write
ls -la $out
