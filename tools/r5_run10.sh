#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export UA2_LIB=$PWD/tools/ubench/dbg/libua2hip_g2stamp.so
(timeout 120 python tools/ubench/g2_stamps.py 6272 5120 3072 16; timeout 120 python tools/ubench/g2_stamps.py 6272 8192 3072 8; timeout 120 python tools/ubench/g2_stamps.py 1000 4608 1536 8) > gpurun_out/r5_g2_stamps.txt 2>&1
grep -v amdgpu gpurun_out/r5_g2_stamps.txt
