#!/bin/bash
# attention micro-benchmark over the variant libraries under tools/variants/ (two alternating rounds on one box)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in "$@"; do
    echo "== $v"; MMSIM_LIB=$PWD/tools/variants/$v.so python tools/bench_attn.py 2>&1 | grep dropout
  done
done
