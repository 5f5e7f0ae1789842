#!/bin/bash
# interleaved A/B of several variant libraries on the secondary configs (run from the repo root on the GPU box):
#   ab_multi.sh "<name1> <name2> ..." <reps> [bench_configs args]     names = scratch/lib_<name>/ (build_variant.sh); "main" = the tree's library
names=$1; reps=$2; shift 2
for rep in $(seq 1 $reps); do
  for n in $names; do
    if [ "$n" = main ]; then unset GS360_LIB; else export GS360_LIB=$PWD/scratch/lib_$n/libgs360hip.so; fi
    echo "== $n rep $rep"
    python tests/tools/bench_configs.py --steps 30 "$@" | python3 -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('   ', d['config'][:60].ljust(60), d.get('us_per_frame', d.get('ms_per_pair', d.get('ms_per_image'))), d['frac_of_8TBps'], d['parity_vs_oracle'])"
  done
done
