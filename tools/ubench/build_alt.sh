#!/bin/bash
# Experiment builds of libua2hip.so: one source file recompiled with extra -D flags, the other objects taken from the
# production build (uniaudio2_amd/build/*.o).  Usage: build_alt.sh <tag> <file.hip> [-DX=Y ...]  ->  tools/ubench/dbg/libua2hip_<tag>.so
# (load with UA2_LIB=<path>).
set -e
cd "$(dirname "$0")/../.."
tag=$1; src=$2; shift 2
mkdir -p tools/ubench/dbg
base=$(basename "$src")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c uniaudio2_amd/csrc/$base -o tools/ubench/dbg/${base}_$tag.o
objs=$(ls uniaudio2_amd/build/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/ubench/dbg/${base}_$tag.o -o tools/ubench/dbg/libua2hip_$tag.so
echo tools/ubench/dbg/libua2hip_$tag.so
