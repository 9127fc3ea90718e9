# L2 hit rate + fabric read requests per kernel (one factorization, no look-ahead so kernels run alone).
# usage: bash tools/pmc_l2.sh <tag> [bench args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
MPQR_TPOLL=0 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $root/gpurun_out/pmc_l2_$tag -o l2 --output-format csv -- python3 $root/bench.py --no-cpu-baseline --no-dropin --no-alone --steps 1 --warmup 0 --no-lookahead "$@" > $root/gpurun_out/pmc_l2_$tag.log 2>&1
cd $root
python3 - "$(find gpurun_out/pmc_l2_$tag -name "l2_counter_collection.csv" | head -1)" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.Counter()); calls = collections.Counter(); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('mpqr::', '').replace('void ', '')[:48]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen: seen.add(r['Dispatch_Id']); calls[n]+=1
print("%-50s %5s %12s %12s %8s %14s" % ("kernel","calls","TCC_HIT","TCC_MISS","hit%","EA0_RDREQ(64B)"))
for n, c in sorted(agg.items(), key=lambda kv: -(kv[1]['TCC_HIT_sum']+kv[1]['TCC_MISS_sum']))[:10]:
    h, m = c['TCC_HIT_sum'], c['TCC_MISS_sum']
    print("%-50s %5d %12.4g %12.4g %8.1f %14.4g" % (n, calls[n], h, m, 100*h/max(h+m,1), c['TCC_EA0_RDREQ_sum']))
PY
rm -rf gpurun_out/pmc_l2_$tag
