# rocprofv3 kernel traces of bench.py kept under profiles/ (run on the GPU box: bash tools/profile_round.sh r02_a)
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-r02}
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_two -- python3 bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-graph > gpurun_out/${tag}_two.log 2>&1
MMSIM_TWO_STREAMS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_one -- python3 bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-graph > gpurun_out/${tag}_one.log 2>&1
# steps in each trace: two-stream run = 3 warm-up + 7 timed + 3 one-stream steps bench.py appends; one-stream run = 3 + 7
for v in two one; do
  db=$(find gpurun_out/prof_${tag}_$v -name "*results.db" | head -1)
  if [ $v = two ]; then n=13; else n=10; fi
  python tools/prof_summary.py $db $n gpurun_out/${tag}_${v}_kernel_stats.csv 12
  grep -h '"metric"' gpurun_out/${tag}_$v.log | tail -1 > gpurun_out/${tag}_${v}_bench.json
done
