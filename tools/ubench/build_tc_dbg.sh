#!/bin/bash
# Timing-only builds of libua2hip.so with one phase of the pipelined tc conv kernel knocked out (UA2_TC_DBG bits, see
# csrc/ua2_convtc.hip): tools/ubench/dbg/libua2hip_dbg<N>.so; load with UA2_LIB=<path>.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ubench/dbg
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DUA2_TC_DBG=$n -c uniaudio2_amd/csrc/ua2_convtc.hip -o tools/ubench/dbg/convtc_$n.o
  objs=$(ls uniaudio2_amd/build/*.o | grep -v ua2_convtc)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/ubench/dbg/convtc_$n.o -o tools/ubench/dbg/libua2hip_dbg$n.so
done
