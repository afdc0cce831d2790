#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_gemm2.py tests/test_gpu_lm.py tests/test_gpu_configs.py tests/test_gpu_invariance.py -m gpu -x -q -s 2>&1 | grep -v "File \"/usr" | tail -30 > gpurun_out/r5_run7_tests.txt
timeout 300 python tools/ubench/dit_diag.py > gpurun_out/r5_dit_diag_auto.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stage2 -- python $R/tools/ubench/prof_legs.py stage2 > $R/gpurun_out/r5_stage2_leg_prof.txt 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_stage2 -name "*.db" | head -1) > gpurun_out/r5_stage2_kernel_stats.txt 2>/dev/null
rm -rf gpurun_out/prof_stage2
tail -12 gpurun_out/r5_run7_tests.txt | cut -c1-300; grep -v amdgpu gpurun_out/r5_dit_diag_auto.txt | tail -5; head -30 gpurun_out/r5_stage2_kernel_stats.txt | cut -c1-220
