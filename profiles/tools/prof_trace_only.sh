#!/bin/bash
# kernel trace of the headline command alone (the same command as prof_r06_final.sh's first step) with a plain bench line before and after it:
# boxes of this pool differ by ~10 % in launch time; run on several, keep each box's files (gpurun_out/trace_only_<tag>/)
R=$PWD; TAG=${1:-a}; OUT=$R/gpurun_out/trace_only_$TAG; rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/bench_before.json 2> $OUT/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-memsys > $OUT/bench_under_rocprof.json 2>$OUT/trace.err
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/trace
cd $R
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/bench_after.json 2>> $OUT/err.txt
[ -x scratch/membench ] && (scratch/membench rows 4 2 8 15; scratch/membench rowsmix 4 2 8 15 1; scratch/membench dmamix 4 2 8 15 1) > $OUT/membench.txt 2>&1
python - $OUT <<'PY'
import json, sys, csv
O = sys.argv[1]
for f in ("bench_before.json", "bench_under_rocprof.json", "bench_after.json"):
    d = json.loads(open(f"{O}/{f}").read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, d["value"], r["frac"], r["kernel_ms"], r["memsys"]["frac_of_mix"], d["config"]["clocks"]["during"].get("sclk_mhz"))
for row in csv.DictReader(open(f"{O}/kernel_stats.csv")):
    if "eq_srcmajor_kernel" in row["Name"]: print("rocprofv3:", row["Calls"], "launches, average", row["AverageNs"], "ns")
print(open(f"{O}/membench.txt").read())
PY
