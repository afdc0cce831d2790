#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -s 2>&1 | grep -v "File \"/usr" | tail -60 > gpurun_out/r5_full_gpu_suite.txt
grep -E "passed|failed|FAILED|rms|order-free|Error" gpurun_out/r5_full_gpu_suite.txt | tail -30 | cut -c1-300
