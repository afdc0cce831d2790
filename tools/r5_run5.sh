#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_codec_model.py tests/test_gpu_cli.py tests/test_gpu_codec.py tests/test_tokenizer_golden.py -m gpu -x -q -s 2>&1 | grep -v "File \"/usr" | tail -40 > gpurun_out/r5_stage2_tests.txt
tail -30 gpurun_out/r5_stage2_tests.txt | cut -c1-250
