#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_attn.py tests/test_gpu_codec_model.py tests/test_gpu_invariance.py -m gpu -x -q -s 2>&1 | grep -v "File \"/usr" | tail -25 > gpurun_out/r5_run12_tests.txt
timeout 300 python tools/ubench/dit_diag.py > gpurun_out/r5_dit_diag_attn.txt 2>&1
UA2_DIT_ATTN_SPLIT=1 timeout 300 python tools/ubench/dit_diag.py > gpurun_out/r5_dit_diag_attn_split.txt 2>&1
timeout 600 python tools/ubench/prof_legs.py stage2 > gpurun_out/r5_stage2_leg_v3.txt 2>&1
grep -E "passed|failed|FAILED|flash|batched vs|DiT torch" gpurun_out/r5_run12_tests.txt | tail -12 | cut -c1-250
for f in gpurun_out/r5_dit_diag_attn.txt gpurun_out/r5_dit_diag_attn_split.txt; do echo "== $f"; grep -v amdgpu $f | sed -n 4,7p; done
grep -v "amdgpu\|Warn\|warn" gpurun_out/r5_stage2_leg_v3.txt | tail -2 | cut -c1-2500
