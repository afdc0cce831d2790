#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/ubench/dit_diag.py > gpurun_out/r5_dit_diag_deep.txt 2>&1
UA2_GEMM2_NO_DEEP=1 timeout 300 python tools/ubench/dit_diag.py > gpurun_out/r5_dit_diag_nodeep.txt 2>&1
UA2_GEMM2_DEEP_MAX_GRID=100000 timeout 300 python tools/ubench/dit_diag.py > gpurun_out/r5_dit_diag_alldeep.txt 2>&1
timeout 900 python -X faulthandler -m pytest tests/test_gpu_gemm2.py tests/test_gpu_configs.py -m gpu -x -q -s 2>&1 | grep -v "File \"/usr" | tail -12 > gpurun_out/r5_run8_tests.txt
for f in gpurun_out/r5_dit_diag_deep.txt gpurun_out/r5_dit_diag_nodeep.txt gpurun_out/r5_dit_diag_alldeep.txt; do echo "== $f"; grep -v amdgpu $f | tail -4; done; tail -8 gpurun_out/r5_run8_tests.txt | cut -c1-250
