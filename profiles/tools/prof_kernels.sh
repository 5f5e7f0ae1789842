#!/bin/bash
# Per-kernel rocprofv3 evidence for the kernels other than the cfg2 headline (run from the repo root on the GPU box):
#   prof_kernels.sh <outdir-under-gpurun_out> <label> <bench_configs args...>
# One --kernel-trace --stats pass plus separate --pmc passes (never combined), each under its own timeout; prints and
# stores per-kernel means.
cd /tmp && export TMPDIR=/tmp
R=/root/repo; OUT=$R/gpurun_out/$1; LABEL=$2; shift 2
mkdir -p $OUT/$LABEL; O=$OUT/$LABEL
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tests/tools/bench_configs.py "$@" > $O/bench.json 2>$O/trace.err
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tests/tools/bench_configs.py "$@" > $O/b$i.log 2>&1 || echo "pass $i failed/timeout"
done
python3 - $O <<'PY'
import csv,glob,collections,json,sys
O=sys.argv[1]
out=collections.defaultdict(dict)
for f in sorted(glob.glob(O+'/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)): acc[(r['Kernel_Name'].split('(')[0],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in acc.items(): out[k][c]={'mean_per_launch':sum(v)/len(v),'launches':len(v)}
json.dump(out,open(O+'/pmc_summary.json','w'),indent=1)
for k,d in out.items():
    print(k)
    for c,v in d.items(): print('   ',c,round(v['mean_per_launch']))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
cat $O/kernel_stats.csv | cut -c1-200
cat $O/bench.json
