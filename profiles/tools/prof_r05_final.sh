#!/bin/bash
# round-5 final measurements on the shipped library (run from the repo root on the GPU box): headline kernel trace + PMC passes (the
# headline now launches eq_srcmajor_kernel), the plain bench line (with its `secondary` array), the full secondary table, the 600-frame
# resident job, a fuzz campaign over the kernel-selection options (context options: the library reads no environment variable).
R=$PWD; OUT=$R/gpurun_out/prof_r05_final; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2>$OUT/trace.err
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TD_TD_BUSY_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/b$i.log 2>&1 || echo "pass $i failed/timeout"
done
python3 - $OUT <<'PY'
import csv,glob,collections,json,sys
O=sys.argv[1]; out={}
for f in sorted(glob.glob(O+'/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "eq_srcmajor_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): out[k]={'mean_per_launch':sum(v)/len(v),'launches':len(v)}
json.dump(out,open(O+'/pmc_summary.json','w'),indent=1)
for k,v in out.items(): print(k,round(v['mean_per_launch']))
if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
    hb=int((2*out['FETCH_SIZE']['mean_per_launch']+out['WRITE_SIZE']['mean_per_launch'])*1024)
    json.dump({"kernel":"eq_srcmajor_kernel","frames_per_launch":16,"hbm_bytes_per_launch":hb,
               "formula":"(2*FETCH_SIZE + WRITE_SIZE) KB * 1024 (gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads -- the tile copies are 16 B per lane; calibrated in round 1, profiles/README.md)",
               "source":"profiles/r05/final_pmc_summary.json (2 x FETCH_SIZE + WRITE_SIZE, KB x 1024)"},open(O+'/hbm_traffic.json','w'),indent=1)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; rm -rf $OUT/p*/ $OUT/trace
cd $R
python bench.py --steps 100 --warmup 10 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_settings.json 2>> $OUT/bench_plain.err
python tests/tools/bench_configs.py --steps 40 > $OUT/bench_configs.jsonl 2> $OUT/bench_configs.err
python bench.py --mode job > $OUT/bench_job600_1gpu.json 2> $OUT/bench_job.err
python profiles/tools/srcmajor/sweep_capi.py > $OUT/srcmajor_tile_sweep.txt 2>&1
python profiles/tools/srcmajor/sweep_rings.py > $OUT/srcmajor_other_rings.txt 2>&1
python profiles/tools/srcmajor/sweep_frames.py > $OUT/srcmajor_frames_sweep.txt 2>&1
python profiles/tools/srcmajor/sweep_families.py 4 16 > $OUT/srcmajor_family_sweep.txt 2>&1
(for spec in "" "lanemap=0" "lanemap=1" "ring=1" "lanemap=0 stage=1" "lanemap=0 stage=1 ring=2" "srcmajor=1" "srcmajor=1 --only srcmajor"; do
   args=""; for o in $spec; do case $o in --only) args="$args --only";; srcmajor) args="$args srcmajor";; *=*) args="$args --option $o";; esac; done
   echo "== options: ${spec:-defaults}"; timeout 400 python tests/tools/fuzz_parity.py --seconds 90 --seed 5$RANDOM $args | tail -1; done) > $OUT/fuzz_campaign.txt 2>&1
bash profiles/tools/srcmajor/pmc_cfg.sh cfg3 1 > $OUT/pmc_cfg3_srcmajor.txt 2>&1
bash profiles/tools/srcmajor/pmc_cfg.sh cfg3 0 > $OUT/pmc_cfg3_staged.txt 2>&1
cat $OUT/kernel_stats.csv | cut -c1-160; head -c 600 $OUT/bench_plain.json; echo; cat $OUT/fuzz_campaign.txt
