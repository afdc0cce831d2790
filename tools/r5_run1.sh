#!/bin/bash
# round 5, first GPU call: the order-free GEMM — parity tests, shape table, the DiT step in situ
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm2.py -x -q 2>&1 | tail -40 > gpurun_out/r5_gemm2_tests.txt
timeout 400 python tools/ubench/gemm2_shapes.py > gpurun_out/r5_gemm2_shapes.txt 2>&1
timeout 300 python tools/ubench/dit_diag.py > gpurun_out/r5_dit_diag_free.txt 2>&1
UA2_DIT_SUM_ORDER=0 timeout 300 python tools/ubench/dit_diag.py > gpurun_out/r5_dit_diag_inv.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_codec_model.py -x -q 2>&1 | tail -15 > gpurun_out/r5_codec_model_tests.txt
tail -5 gpurun_out/r5_gemm2_tests.txt; cat gpurun_out/r5_gemm2_shapes.txt; tail -4 gpurun_out/r5_dit_diag_free.txt gpurun_out/r5_dit_diag_inv.txt; tail -3 gpurun_out/r5_codec_model_tests.txt
