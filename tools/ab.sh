#!/bin/bash
# A/B two builds of the library on one box: alternating bench runs, ms_per_step of each.  usage: tools/ab.sh <base.so> <rounds>
base=$1; rounds=${2:-2}
for i in $(seq $rounds); do
  MMSIM_LIB=$base python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; print("base ms_per_step", json.loads(sys.stdin.read())["ms_per_step"])'
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; print("new  ms_per_step", json.loads(sys.stdin.read())["ms_per_step"])'
done
