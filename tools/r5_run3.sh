#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_codec_model.py tests/test_gpu_cli.py tests/test_gpu_codec.py tests/test_tokenizer_golden.py -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r5_stage2_tests.txt
timeout 600 python tools/ubench/prof_legs.py stage2 > gpurun_out/r5_stage2_leg.txt 2>&1
tail -25 gpurun_out/r5_stage2_tests.txt; grep -v amdgpu.ids gpurun_out/r5_stage2_leg.txt | tail -5 | cut -c1-3000
