#!/bin/bash
# Build a second copy of the library with extra -D flags for ONE source file (A/B on one box with MMSIM_LIB):
#   tools/build_variant.sh <out.so> <file.hip> [-DNAME=VALUE ...]
# All other objects are taken from multimodalsimilar_amd/build/ (run `python -m multimodalsimilar_amd.build` first).
set -e
out=$1; src=$2; shift 2
here=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/variant_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -ffp-contract=fast "$@" -c $here/multimodalsimilar_amd/csrc/$src -o $obj
objs=""
for f in core gemm gemm_fast attention rowwise head_optim conv mbconv dwmfma search preprocess; do
  if [ "$f.hip" = "$src" ]; then objs="$objs $obj"; else objs="$objs $here/multimodalsimilar_amd/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
rm -f $obj
echo built $out
