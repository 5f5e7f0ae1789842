#!/bin/bash
for n in "$@"; do
  IFS=@ read name vars <<< "$n"
  if [ "$name" = product ]; then lib=360cam-pgm-3dgs-tools_amd/lib/libgs360hip.so; else lib=scratch/lib_$name/libgs360hip.so; fi
  echo "== $name $vars"
  GS360_LIB=$lib timeout 120 python tests/tools/bench_cfg4_stage.py --steps 30 --variants $vars 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(r['ms_per_pair'], r['parity_vs_oracle'], r['table_stage'], r['rows'], r['wgs'], r.get('loaders'), r['staged_jobs'])
"
done
