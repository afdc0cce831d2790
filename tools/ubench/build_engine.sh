#!/bin/bash
# builds the persistent-engine prototype (tools/ubench/engine.hip) next to its harness; not part of libua2hip.so
cd "$(dirname "$0")/../.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
  -I include -I uniaudio2_amd/csrc "$@" tools/ubench/engine.hip -o tools/ubench/libengine.so
