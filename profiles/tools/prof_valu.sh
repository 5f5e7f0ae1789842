#!/bin/bash
# Vector-ALU busy fraction of the CUBIC kernels (the reference tools' default interpolation: PC:730, DF:229-234): they are arithmetic-bound,
# an HBM fraction is the wrong yardstick for them.  One PMC pass per config (SQ + GRBM counters fit one pass), per-kernel means ->
#   valu_busy = SQ_ACTIVE_INST_VALU x 4 / (SIMDs x launch cycles),  launch cycles = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs), 1024 SIMDs
# (the gfx94x VALUBusy formula of rocprof's derived_counters.xml, which ROCm 7.2 falls back to on gfx950: MI355X_MICROARCH.md).
# Writes gpurun_out/<out>/valu_busy.json; copy it to profiles/valu_busy.json (bench.py attaches it to the cubic `secondary` rows).
R=$PWD; OUT=$R/gpurun_out/${1:-prof_valu}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SET="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"
run() { # key, script args...
  key=$1; shift
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/p_$key -- python3 "$@" > $OUT/$key.log 2>&1 || echo "$key failed/timeout"
}
run cfg2-cubic $R/tests/tools/bench_configs.py --steps 20 --only equirect --eq cfg2cubic
run cfg1-cubic $R/tests/tools/bench_configs.py --steps 20 --only equirect --eq cfg1cubic
run cfg3-cubic $R/tests/tools/bench_configs.py --steps 20 --only equirect --eq cfg3cubic
run cfg4-cubic-plans $R/tests/tools/bench_cfg4_stage.py --steps 20 --interp 2 --variants 0:32:0
run cfg4-linear-plans $R/tests/tools/bench_cfg4_stage.py --steps 20 --interp 1 --variants -1:32:0
run cfg5 $R/tests/tools/bench_configs.py --steps 20 --only equirect --eq cfg5
python3 - $OUT <<'PY'
import csv, glob, collections, json, os, sys
O = sys.argv[1]; out = {}
for d in sorted(glob.glob(O + '/p_*')):
    key = os.path.basename(d)[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    best = None
    for k, c in acc.items():                               # the config's own kernel: the one with the most vector-ALU work per launch
        if 'SQ_ACTIVE_INST_VALU' not in c or 'plan' in k or 'pack' in k: continue
        tot = sum(c['SQ_ACTIVE_INST_VALU'])
        if best is None or tot > best[1]: best = (k, tot)
    if not best: continue
    c = acc[best[0]]
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m['GRBM_GUI_ACTIVE'] / 8.0
    out[key] = {"kernel": best[0].replace('void gs360::', ''), "launches": len(c['SQ_ACTIVE_INST_VALU']), "launch_cycles": round(cyc),
                "valu_instructions": round(m['SQ_INSTS_VALU']), "valu_busy": round(m['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc), 3),
                "lds_instructions": round(m.get('SQ_INSTS_LDS', 0)),
                "wave_cycles_split": {"active": round(m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'], 3), "parked": round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 3),
                                      "issue_stall": round(m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], 3)}}
json.dump({"formula": "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) per launch, rocprofv3 --pmc (one pass per config): profiles/tools/prof_valu.sh",
           "configs": out}, open(O + '/valu_busy.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*counter_collection.csv" -delete; rm -rf $OUT/p_*
