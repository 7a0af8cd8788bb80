#!/bin/bash
# A/B an environment switch on one box: alternating bench runs.  usage: tools/ab_env.sh VAR=a VAR=b [rounds]
a=$1; b=$2; rounds=${3:-2}
for i in $(seq $rounds); do
  env $a python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; print(sys.argv[1], "ms_per_step", json.loads(sys.stdin.read())["ms_per_step"])' $a
  env $b python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; print(sys.argv[1], "ms_per_step", json.loads(sys.stdin.read())["ms_per_step"])' $b
done
