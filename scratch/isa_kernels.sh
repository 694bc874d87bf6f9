#!/bin/bash
mkdir -p /tmp/st
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-slp-vectorize -Iinclude -Iaidadsp-lv2_amd/csrc $EXTRA \
   -S --cuda-device-only -o /tmp/st/kernels.s aidadsp-lv2_amd/csrc/aidax_kernels.hip
