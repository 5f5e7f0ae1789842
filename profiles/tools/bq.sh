#!/bin/bash
# usage: bq.sh <label> <frames> [extra bench args]   -> prints label, frames, kernel_ms, us/frame, frac
label=$1; frames=$2; shift 2
python /root/repo/bench.py --steps 100 --warmup 10 --no-cpu-baseline --frames $frames "$@" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; f=d['config']['frames_per_step']
print('$label', 'frames', f, 'kernel_ms', r['kernel_ms'], 'us/frame', round(r['kernel_ms']/f*1000,2), 'frac', r['frac'])"
