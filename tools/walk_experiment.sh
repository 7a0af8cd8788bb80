set -e
cd $GRAFT_REPO_ROOT
for b in 8 -6 -4 -3; do
  echo "== band $b"; MMSIM_GEMM_BAND=$b timeout -k 10 200 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
done
export TMPDIR=/tmp
for b in 8 -6 -4; do
  MMSIM_GEMM_BAND=$b timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f_$b -- python3 tools/pmc_gemm.py > gpurun_out/pmc_f_$b.log 2>&1
  echo "== fetch band $b"; python tools/pmc_parse.py gpurun_out/pmc_f_$b FETCH_SIZE
done
