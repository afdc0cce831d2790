#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_gemm2.py tests/test_gpu_lm.py tests/test_gpu_configs.py -m gpu -x -q -s 2>&1 | grep -v "File \"/usr" | tail -30 > gpurun_out/r5_run6_tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-legs > gpurun_out/r5_run6_bench.json 2> gpurun_out/r5_run6_bench.err
UA2_ORDER_FREE_ROWS=2048 timeout 300 python tools/ubench/prof_legs.py config3 > gpurun_out/r5_config3_free.txt 2>&1
UA2_ORDER_FREE_ROWS=1024 timeout 300 python tools/ubench/prof_legs.py batched1024 > gpurun_out/r5_b1024_free.txt 2>&1
tail -12 gpurun_out/r5_run6_tests.txt | cut -c1-300; cat gpurun_out/r5_run6_bench.json | cut -c1-2500; tail -3 gpurun_out/r5_run6_bench.err
for f in gpurun_out/r5_config3_free.txt gpurun_out/r5_b1024_free.txt; do grep -v "amdgpu.ids" $f | tail -2 | cut -c1-700; done
