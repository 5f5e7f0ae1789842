#!/bin/bash
# LDS-staging feasibility study for eq_views_kernel (cfg2), run from the repo root on the GPU box.
#   1. export_plan.py (CPU, oracle map): the exact per-(tile, pass) 128-B line lists a staged kernel would load -> cfg2_plan.bin
#   2. ldsdma_probe: synthetic cold 8-line runs -> LDS-DMA vs register-staged vs plain loads vs scalar-cache loads
#   3. replay_probe: replays cfg2_plan.bin (8 frames/launch, XCD-chunked) = staging-only time of a staged kernel
# rocprofv3: one --kernel-trace --stats pass and separate --pmc passes per variant.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$R/profiles/tools/lds_staging; OUT=$R/gpurun_out/lds_staging; mkdir -p $OUT
cd $T
hipcc --offload-arch=gfx950 -O3 -o $OUT/ldsdma_probe ldsdma_probe.hip 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -o $OUT/replay_probe replay_probe.hip 2>/dev/null
[ -f $OUT/cfg2_plan.bin ] || python3 export_plan.py $OUT/cfg2_plan.bin > $OUT/export.log 2>&1
{ for a in "20 8 2 0" "40 8 1 0" "40 8 1 60"; do $OUT/ldsdma_probe $a; done; } > $OUT/ldsdma_probe.txt 2>&1
{ for a in "512 0 0" "384 0 0" "256 0 0" "384 1 0" "512 0 2" "512 1 2"; do $OUT/replay_probe $OUT/cfg2_plan.bin $a; done; } > $OUT/replay_probe.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for v in "384 1 0" "512 0 2"; do
  set -- $v; tag=cap$1_order$2_mode$3
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/trace -- $OUT/replay_probe $OUT/cfg2_plan.bin $1 $2 $3 20 > /dev/null 2>&1
  cp $OUT/$tag/trace/*/*kernel_stats.csv $OUT/${tag}_kernel_stats.csv
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag/p$i -- $OUT/replay_probe $OUT/cfg2_plan.bin $1 $2 $3 5 > /dev/null 2>&1 || echo "pass $i failed"
  done
  python3 - $OUT/$tag $OUT/${tag}_pmc.json <<'PY'
import csv,glob,collections,json,sys
out={}
for f in sorted(glob.glob(sys.argv[1]+'/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'replay' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): out[k]={'mean_per_launch':sum(v)/len(v),'launches':len(v)}
json.dump(out,open(sys.argv[2],'w'),indent=1)
print(sys.argv[2],{k:round(v['mean_per_launch']) for k,v in out.items()})
PY
  rm -rf $OUT/$tag
done
cat $OUT/ldsdma_probe.txt $OUT/replay_probe.txt
