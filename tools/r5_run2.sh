#!/bin/bash
# round 5, second GPU call: variants / knock-outs of the order-free GEMM, the LM legs with it forced on (in situ)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ubench/gemm2_variants.py > gpurun_out/r5_gemm2_variants.txt 2>&1
for d in 1 2 4 5; do
  UA2_LIB=$PWD/tools/ubench/dbg/libua2hip_g2dbg$d.so UA2_VARIANTS=b16,b8 UA2_ONLY=swiglu,qkv,dit-qkv timeout 200 python tools/ubench/gemm2_variants.py > gpurun_out/r5_gemm2_knock_$d.txt 2>&1
done
timeout 300 python tools/ubench/prof_legs.py config3 > gpurun_out/r5_config3_inv.txt 2>&1
UA2_GEMM2_FORCE=1 UA2_GEMM2_MIN_ROWS=2048 timeout 300 python tools/ubench/prof_legs.py config3 > gpurun_out/r5_config3_free.txt 2>&1
timeout 300 python tools/ubench/prof_legs.py batched1024 > gpurun_out/r5_b1024_inv.txt 2>&1
UA2_GEMM2_FORCE=1 UA2_GEMM2_MIN_ROWS=1024 timeout 300 python tools/ubench/prof_legs.py batched1024 > gpurun_out/r5_b1024_free.txt 2>&1
cat gpurun_out/r5_gemm2_variants.txt | grep -v amdgpu.ids
for f in gpurun_out/r5_gemm2_knock_*.txt gpurun_out/r5_config3_*.txt gpurun_out/r5_b1024_*.txt; do echo "== $f"; grep -v amdgpu.ids $f | tail -4 | cut -c1-400; done
