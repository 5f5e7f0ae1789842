#!/bin/bash
# PMC evidence for one equirect config on one library (run from the repo root on the GPU box):
#   prof_pmc.sh <outdir-under-gpurun_out> <label> <lib-name|main> <bench_configs args...>
# One --kernel-trace --stats pass plus separate --pmc passes (never combined with a trace), each under its own timeout; stores the
# per-kernel means as pmc_summary.json.  <lib-name> = scratch/lib_<name>/ (profiles/tools/build_variant.sh).
R=$PWD; OUT=$R/gpurun_out/$1; LABEL=$2; LIB=$3; shift 3
if [ "$LIB" != main ]; then export GS360_LIB=$R/scratch/lib_$LIB/libgs360hip.so; fi
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT/$LABEL; O=$OUT/$LABEL
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tests/tools/bench_configs.py "$@" > $O/bench.json 2>$O/trace.err
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum TCP_TOTAL_ACCESSES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TD_TD_BUSY_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU2"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tests/tools/bench_configs.py "$@" > $O/b$i.log 2>&1 || echo "pass $i ($set) failed/timeout"
done
python3 - $O <<'PY'
import csv,glob,collections,json,sys
O=sys.argv[1]
out=collections.defaultdict(dict)
for f in sorted(glob.glob(O+'/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)): acc[(r['Kernel_Name'].split('(gs360::')[0].split('(float')[0],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in acc.items(): out[k][c]={'mean_per_launch':sum(v)/len(v),'launches':len(v)}
json.dump(out,open(O+'/pmc_summary.json','w'),indent=1)
for k,d in out.items():
    if 'eq_views' not in k and 'remap' not in k and 'color' not in k and 'fe_views' not in k: continue
    print(k)
    for c,v in d.items(): print('   ',c,round(v['mean_per_launch']))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; rm -rf $O/p*/ $O/trace
cut -c1-160 $O/kernel_stats.csv | head -5
cat $O/bench.json
