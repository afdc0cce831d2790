#!/bin/bash
# the persistent-engine prototype's table (profiles/r5_engine_prototype.txt)
mkdir -p gpurun_out/r15
O=gpurun_out/r15/engine_prototype.txt
{
echo "# tools/ubench/engine_run.py: one depth-decoder pass at B = 1 (17 ops, 536.9 MB bf16), chain of ua2_linear launches vs ONE persistent launch"
echo "## production order (1 loader wave, 3 consumer waves, 2 + 1 gather waves per CU), bit-exact check + per-op stamps"
timeout 120 python tools/ubench/engine_run.py --iters 100 --stamps 2>&1 | grep -v "amdgpu.ids" | grep -v "bit-identical"
echo "## repeat, no stamps"
timeout 120 python tools/ubench/engine_run.py --iters 200 2>&1 | tail -3
echo "## loader thinned to one slot in flight while a gather wave sweeps (flags 1)"
timeout 120 python tools/ubench/engine_run.py --iters 200 --flags 1 2>&1 | tail -1
echo "## two loader waves (96 KiB in flight per CU)"
ENG_LIB=libengine_l2.so timeout 120 python tools/ubench/engine_run.py --iters 200 2>&1 | tail -1
echo "## one layer only (4 ops, 121.6 MB: the guide's launches-baseline workload)"
timeout 120 python tools/ubench/engine_run.py --iters 200 --layers 1 --no-head 2>&1 | tail -3
echo "## timing knock-outs (wrong results by construction)"
for f in 14 76 504 100 112 48 16 32; do timeout 120 python tools/ubench/engine_run.py --iters 100 --flags $f 2>&1 | tail -1; done
echo "## the same knock-outs, two loader waves"
for f in 76 504 100 112; do ENG_LIB=libengine_l2.so timeout 120 python tools/ubench/engine_run.py --iters 100 --flags $f 2>&1 | tail -1; done
} > $O 2>&1
cat $O | cut -c1-200
