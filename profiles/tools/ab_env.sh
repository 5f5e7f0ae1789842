#!/bin/bash
# ab_env.sh "<VAR=val;...|-> ..." reps args : A/B by environment variable on the main library
specs=$1; reps=$2; shift 2
for rep in $(seq 1 $reps); do
  for sp in $specs; do
    echo "== [$sp] rep $rep"
    ( if [ "$sp" != "-" ]; then export ${sp//;/ }; fi
    python tests/tools/bench_configs.py --steps 30 "$@" | python3 -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('   ', d['config'][:60].ljust(60), d.get('us_per_frame', d.get('ms_per_pair', d.get('ms_per_image'))), d['frac_of_8TBps'], d['parity_vs_oracle'])" )
  done
done
