#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python tools/ubench/stage2_debug.py 2>&1 | grep -v "File \"/usr" | tail -30) > gpurun_out/r5_s2dbg_graphs.txt
(UA2_EULER_NO_GRAPH=1 UA2_CODEC_NO_GRAPH=1 timeout 300 python tools/ubench/stage2_debug.py 2>&1 | grep -v "File \"/usr" | tail -30) > gpurun_out/r5_s2dbg_nographs.txt
(UA2_CODEC_NO_GRAPH=1 timeout 300 python tools/ubench/stage2_debug.py 2>&1 | grep -v "File \"/usr" | tail -30) > gpurun_out/r5_s2dbg_eulergraph.txt
(UA2_EULER_NO_GRAPH=1 timeout 300 python tools/ubench/stage2_debug.py 2>&1 | grep -v "File \"/usr" | tail -30) > gpurun_out/r5_s2dbg_codecgraph.txt
for f in gpurun_out/r5_s2dbg_*.txt; do echo "== $f"; grep -v "amdgpu.ids\|WeightNorm\|weight_norm" $f | tail -22 | cut -c1-200; done
