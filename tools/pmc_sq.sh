# SQ stall breakdown per kernel (one factorization).  usage: bash tools/pmc_sq.sh [config]
cfg=${1:-c4}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
MPQR_TPOLL=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $root/gpurun_out/pmc_sq -o $cfg --output-format csv -- python3 $root/bench.py --config $cfg --no-cpu-baseline --no-dropin --no-alone --steps 1 --warmup 0 > $root/gpurun_out/pmc_sq_$cfg.log 2>&1
cd $root
python3 - "$(find gpurun_out/pmc_sq -name "${cfg}_counter_collection.csv" | head -1)" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.Counter()); calls = collections.Counter(); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('mpqr::', '').replace('void ', '')[:40]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    key=(r['Dispatch_Id']);
    if key not in seen: seen.add(key); calls[n]+=1
names = ['SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_VALU_MFMA_BUSY_CYCLES','SQ_WAIT_INST_LDS','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE']
print("%-42s %5s %12s  wait_any inst_stall active | mfma_busy/wave_cyc(x4) lds_stall bankconf/ldsactive" % ("kernel","calls","wave_cyc"))
for n, c in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])[:12]:
    w = c['SQ_WAVE_CYCLES'] or 1
    print("%-42s %5d %12.3g  %6.2f %8.2f %8.2f | %8.3f %10.2f %8.3f" % (n, calls[n], w, c['SQ_WAIT_ANY']/w, c['SQ_WAIT_INST_ANY']/w, c['SQ_ACTIVE_INST_ANY']/w, c['SQ_VALU_MFMA_BUSY_CYCLES']/(4*w), c['SQ_WAIT_INST_LDS']/w, c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1)))
PY
