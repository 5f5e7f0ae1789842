#!/bin/bash
# ab_bench.sh <libA> <libB> [rounds]: bench.py headline (no secondary rows) alternating between two libraries on the same box
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for L in $A $B; do
    if [ "$L" = main ]; then lib=360cam-pgm-3dgs-tools_amd/lib/libgs360hip.so; else lib=scratch/lib_$L/libgs360hip.so; fi
    GS360_LIB=$lib python bench.py --steps 20 --warmup 5 --no-secondary 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); rf = r['roofline']
print('$L', r['value'], 'MPix/s', rf['kernel_ms'], 'ms/launch', 'frac', rf['frac'], rf['kernel'], 'parity', r['config']['parity_vs_oracle'])"
  done
done
