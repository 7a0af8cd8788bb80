#!/bin/bash
# SQ counters (one pass: 8 SQ slots) of the kernels whose name contains $2, over the script $1: where the waves' cycles go
#   tools/pmc_sq.sh tools/bench_dwm.py dwm_ gpurun_out/sq_dwm
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
script=$1; filt=$2; out=${3:-gpurun_out/pmc_sq}
NREP=2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $out -- python3 $script > $out.log 2>&1
python - "$out" "$filt" <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:44] + " grid " + r["Grid_Size"], r["Counter_Name"])
        n, s = acc.get(k, (0, 0.0))
        acc[k] = (n + 1, s + float(r["Counter_Value"]))
for n in sorted(set(k[0] for k in acc)):
    if sys.argv[2] not in n: continue
    d = {c: acc[(n, c)][1] / acc[(n, c)][0] for (nn, c) in acc if nn == n}
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(n, {k.replace("SQ_", ""): f"{v / wc:.2f}" for k, v in d.items() if k != "SQ_WAVE_CYCLES"}, f"wave_cycles {wc:.3g}")
PY
