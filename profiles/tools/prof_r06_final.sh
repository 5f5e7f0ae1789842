#!/bin/bash
# round-6 final measurements on the shipped library (run from the repo root on the GPU box): headline kernel trace + PMC passes, the plain
# bench line (with clocks and the `secondary` array) three times a minute apart (the box drifts between two speeds: section 6 of DESIGN.md),
# the full secondary table, the resident and the host-fed job, cfg4's two table kernels under the counters, a fuzz campaign over the
# kernel-selection options, the whole GPU test suite.
R=$PWD; OUT=$R/gpurun_out/prof_r06_final; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-memsys > $OUT/bench_under_rocprof.json 2>$OUT/trace.err
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TD_TD_BUSY_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-memsys > $OUT/b$i.log 2>&1 || echo "pass $i failed/timeout"
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
               "source":"profiles/r06/final_pmc_summary.json (2 x FETCH_SIZE + WRITE_SIZE, KB x 1024)"},open(O+'/hbm_traffic.json','w'),indent=1)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; rm -rf $OUT/p*/ $OUT/trace
cd $R
python bench.py --steps 100 --warmup 10 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
for k in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-secondary > $OUT/bench_driver_settings_$k.json 2>> $OUT/bench_plain.err; sleep 20; done
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_settings.json 2>> $OUT/bench_plain.err
python tests/tools/bench_configs.py --steps 40 > $OUT/bench_configs.jsonl 2> $OUT/bench_configs.err
# the memory system's own rate for the headline's traffic shape on THIS box (profiles/tools/membench.hip), right after the bench lines
[ -x scratch/membench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scratch/membench profiles/tools/membench.hip
(for m in rows dma; do scratch/membench $m 4 2 8 15; done; for m in rowsmix dmamix; do for wg in 2 4; do scratch/membench $m 4 $wg 8 15 1; done; done; scratch/membench rowsmix 4 2 8 15 2; scratch/membench copy 4 2 4 15; scratch/membench write 4 2 8 15) > $OUT/membench_same_box.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/bench_after_membench.json 2>> $OUT/bench_plain.err
python bench.py --mode job > $OUT/bench_job600_1gpu.json 2> $OUT/bench_job.err
python bench.py --mode stream --no-cpu-baseline > $OUT/bench_stream600_1gpu.json 2>> $OUT/bench_job.err
python tests/tools/plan_build_times.py > $OUT/plan_build_times.txt 2>&1
python tests/tools/bench_cfg4_stage.py --steps 40 --variants 0:32:0,1:32:0,1:32:2,1:16:0 > $OUT/cfg4_stage.txt 2>&1
(for spec in "" "lanemap=0" "lanemap=1" "ring=1" "lanemap=0 stage=1" "srcmajor=1" "srcmajor=1 srcmajor_stage=1" "srcmajor=1 --only srcmajor" "table_stage=1" "table_stage=1 table_stage_rows=16 --only tablestage" "--only tablestage"; do
   args=""; nxt=""; for o in $spec; do if [ "$nxt" = only ]; then args="$args --only $o"; nxt=""; elif [ "$o" = "--only" ]; then nxt=only; else args="$args --option $o"; fi; done
   echo "== options: ${spec:-defaults}"; timeout 400 python tests/tools/fuzz_parity.py --seconds 60 --seed 6$RANDOM $args | tail -1; done) > $OUT/fuzz_campaign.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/pytest_gpu.txt
bash profiles/tools/prof_pmc_cmd.sh prof_r06_final/cfg4 staged main table tests/tools/bench_cfg4_stage.py --steps 30 --variants 1:32:0 > $OUT/pmc_cfg4_staged.txt 2>&1
bash profiles/tools/prof_pmc_cmd.sh prof_r06_final/cfg4 gather main table tests/tools/bench_cfg4_stage.py --steps 30 --variants 0:32:0 > $OUT/pmc_cfg4_gather.txt 2>&1
cat $OUT/kernel_stats.csv | cut -c1-160 | head -8; head -c 700 $OUT/bench_plain.json; echo; cat $OUT/fuzz_campaign.txt; cat $OUT/pytest_gpu.txt
