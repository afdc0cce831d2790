#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_gpu_gemm2.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5_run11_tests.txt
UA2_VARIANTS=inv,auto,b16,b8 timeout 600 python tools/ubench/gemm2_variants.py > gpurun_out/r5_gemm2_variants_v2.txt 2>&1
(UA2_LIB=$PWD/tools/ubench/dbg/libua2hip_g2stamp.so timeout 120 python tools/ubench/g2_stamps.py 6272 5120 3072 16) > gpurun_out/r5_g2_stamps_v2.txt 2>&1
tail -3 gpurun_out/r5_run11_tests.txt; grep -v amdgpu gpurun_out/r5_gemm2_variants_v2.txt; grep -v amdgpu gpurun_out/r5_g2_stamps_v2.txt | head -12
