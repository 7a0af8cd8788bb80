#!/bin/bash
# rocprofv3 counter passes over the four forward text-tower GEMMs (tools/pmc_gemm.py), FETCH_SIZE and WRITE_SIZE in separate passes
# (no trace domain besides --kernel-trace), then gpurun_out/pmc_gemm.json in the layout of profiles/r0N_pmc_gemm_pp64.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_gemm_f -- python3 tools/pmc_gemm.py > gpurun_out/pmc_gemm_f.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_gemm_w -- python3 tools/pmc_gemm.py > gpurun_out/pmc_gemm_w.log 2>&1 &&
python tools/pmc_gemm_report.py gpurun_out/pmc_gemm_f gpurun_out/pmc_gemm_w gpurun_out/pmc_gemm.json
