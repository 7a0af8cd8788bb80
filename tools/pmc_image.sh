#!/bin/bash
# rocprofv3 counter passes over the image tower (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one; no trace domains
# besides --kernel-trace).  Output: gpurun_out/pmc_img_{f,w}/ + per-kernel table gpurun_out/pmc_image.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_img_f -- python3 tools/pmc_image.py > gpurun_out/pmc_img_f.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_img_w -- python3 tools/pmc_image.py > gpurun_out/pmc_img_w.log 2>&1
python tools/pmc_image_parse.py gpurun_out/pmc_img_f gpurun_out/pmc_img_w gpurun_out/pmc_image.json
