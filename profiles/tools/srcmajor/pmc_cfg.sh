#!/bin/bash
# kernel trace + PMC passes of one config through the source-major kernel (or, SM=0, the kernels the library picks otherwise); run from the repo root
# usage: pmc_cfg.sh cfg3|cfg1|cfg2 [srcmajor 0/1]
R=$PWD; C=$1; SM=${2:-1}; OUT=$R/gpurun_out/prof_r05_final/pmc_${C}_sm$SM; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/profiles/tools/srcmajor/cfg_loop.py $C 40 $SM > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/profiles/tools/srcmajor/cfg_loop.py $C 4 $SM > $OUT/b$i.log 2>&1 || echo "pass $i failed/timeout"
done
python3 - $OUT <<'PY'
import csv,glob,collections,json,sys
O=sys.argv[1]; out={}
for f in sorted(glob.glob(O+'/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "eq_srcmajor_kernel" in r["Kernel_Name"] or "eq_views_kernel" in r["Kernel_Name"] or "eq_staged" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): out[k]={'mean_per_launch':sum(v)/len(v),'launches':len(v)}
json.dump(out,open(O+'/pmc_summary.json','w'),indent=1)
for k,v in out.items(): print(k,round(v['mean_per_launch']))
PY
grep -h "srcmajor\|eq_views\|eq_staged" $OUT/kernel_stats.csv | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; rm -rf $OUT/p*/ $OUT/trace
