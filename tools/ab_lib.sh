#!/bin/bash
# A/B a benchmark script between the in-tree library and variant builds under tools/variants/ (alternating, two rounds):
#   tools/ab_lib.sh tools/bench_imgemm2.py gemm_pf2off
cd $GRAFT_REPO_ROOT
script=$1; shift
for r in 1 2; do
  echo "== in-tree (round $r)"; python $script 2>&1 | tail -${TAILN:-8}
  for v in "$@"; do echo "== $v (round $r)"; MMSIM_LIB=$PWD/tools/variants/$v.so python $script 2>&1 | tail -${TAILN:-8}; done
done
