#!/bin/bash
# validation of the last host-side changes: order-free rows on the depth decoder, interleaved draws, graph cache sizes
mkdir -p gpurun_out/r13
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_codec_model.py tests/test_gpu_lm.py tests/test_gpu_gemm2.py -m gpu -x -q > gpurun_out/r13/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r13/pytest.log
UA2_ORDER_FREE_ROWS=1024 timeout 300 python tools/ubench/prof_legs.py batched1024 > gpurun_out/r13/b1024_of.log 2>&1
timeout 300 python tools/ubench/prof_legs.py batched1024 > gpurun_out/r13/b1024.log 2>&1
UA2_ORDER_FREE_ROWS=2048 timeout 300 python tools/ubench/prof_legs.py config3 > gpurun_out/r13/c3_of.log 2>&1
tail -3 gpurun_out/r13/pytest.log; tail -1 gpurun_out/r13/b1024_of.log; tail -1 gpurun_out/r13/b1024.log; tail -1 gpurun_out/r13/c3_of.log
