#!/bin/bash
# SQ counters of the depthwise tile kernels (one pass: 8 SQ slots): where the waves' cycles go
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_dwt -- python3 tools/bench_dwt.py > gpurun_out/pmc_dwt.log 2>&1
python - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("gpurun_out/pmc_dwt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:44] + " grid " + r["Grid_Size"], r["Counter_Name"])
        n, s = acc.get(k, (0, 0.0))
        acc[k] = (n + 1, s + float(r["Counter_Value"]))
names = sorted(set(k[0] for k in acc))
for n in names:
    if "dwt_" not in n: continue
    d = {c: acc[(n, c)][1] / acc[(n, c)][0] for (nn, c) in acc if nn == n}
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(n, {k.replace("SQ_", ""): f"{v / wc:.2f}" for k, v in d.items() if k != "SQ_WAVE_CYCLES"}, f"wave_cycles {wc:.3g}")
PY
