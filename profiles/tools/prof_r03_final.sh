#!/bin/bash
# round-3 final profile of the shipped kernel (run from the repo root on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; OUT=$R/gpurun_out/prof_r03_final; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>$OUT/trace.err
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/b$i.log 2>&1 || echo "pass $i failed/timeout"
done
cat $OUT/trace/*/*kernel_stats.csv
python3 - <<'PY'
import csv,glob,collections,json
out={}
for f in sorted(glob.glob('/root/repo/gpurun_out/prof_r03_final/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "eq_views_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): out[k]={'mean_per_launch':sum(v)/len(v),'launches':len(v)}
json.dump(out,open('/root/repo/gpurun_out/prof_r03_final/pmc_summary.json','w'),indent=1)
for k,v in out.items(): print(k,round(v['mean_per_launch']))
PY
find $OUT -name "*kernel_trace.csv" -delete
cd $R && python bench.py
