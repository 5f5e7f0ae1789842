#!/bin/bash
# runs bench_cfg4_stage for the product lib and each probe lib (parity is expected to fail for the probes)
for n in "" norender nodma nostore nobox; do
  if [ -z "$n" ]; then lib=360cam-pgm-3dgs-tools_amd/lib/libgs360hip.so; else lib=scratch/lib_$n/libgs360hip.so; fi
  echo "== ${n:-product}"
  GS360_LIB=$lib timeout 120 python tests/tools/bench_cfg4_stage.py --steps 30 --variants ${1:-1:32:0} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(r['ms_per_pair'], r['parity_vs_oracle'], r['table_stage'], r['rows'], r['wgs'], r['staged_jobs'])
"
done
