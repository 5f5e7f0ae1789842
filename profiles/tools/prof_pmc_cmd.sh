#!/bin/bash
# PMC evidence for any bench script (run from the repo root on the GPU box):
#   prof_pmc_cmd.sh <outdir-under-gpurun_out> <label> <lib-name|main> <kernel-name-filter> <script.py> <args...>
# One --kernel-trace --stats pass plus separate --pmc passes (never combined with a trace); per-kernel means -> pmc_summary.json.
R=$PWD; OUT=$R/gpurun_out/$1; LABEL=$2; LIB=$3; FILTER=$4; shift 4
if [ "$LIB" != main ]; then export GS360_LIB=$R/scratch/lib_$LIB/libgs360hip.so; fi
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT/$LABEL; O=$OUT/$LABEL
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $SCRIPT "$@" > $O/bench.json 2>$O/trace.err
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $SCRIPT "$@" > $O/b$i.log 2>&1 || echo "pass $i ($set) failed/timeout"
done
python3 - $O "$FILTER" <<'PY'
import csv,glob,collections,json,sys
O,flt=sys.argv[1],sys.argv[2]
out=collections.defaultdict(dict)
for f in sorted(glob.glob(O+'/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)): acc[(r['Kernel_Name'].split('(gs360::')[0].split('(float')[0],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in acc.items(): out[k][c]={'mean_per_launch':sum(v)/len(v),'launches':len(v)}
json.dump(out,open(O+'/pmc_summary.json','w'),indent=1)
for k,d in out.items():
    if flt not in k: continue
    print(k)
    for c,v in d.items(): print('   ',c,round(v['mean_per_launch']), v['launches'])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; rm -rf $O/p*/ $O/trace
grep -i "$FILTER\|Name" $O/kernel_stats.csv | cut -c1-200 | head -8
