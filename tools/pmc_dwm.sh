#!/bin/bash
# kernel-trace + FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only) over tools/bench_dwm.py -> gpurun_out/pmc_dwm.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_dwm}
NREP=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d ${out}_t -- python3 tools/bench_dwm.py > ${out}_t.log 2>&1
NREP=2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${out}_f -- python3 tools/bench_dwm.py > ${out}_f.log 2>&1
NREP=2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${out}_w -- python3 tools/bench_dwm.py > ${out}_w.log 2>&1
python tools/pmc_image_parse.py ${out}_f ${out}_w ${out}.json
db=$(find ${out}_t -name "*results.db" | head -1)
python tools/prof_summary.py $db 1 ${out}_kernel_stats.csv 12
