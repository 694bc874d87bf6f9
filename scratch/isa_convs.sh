#!/bin/bash
# the ISA of aidax_convs.hip's kernels (ship flags) -> /tmp/st/convs.s, and where k_conv_st touches scratch
mkdir -p /tmp/st
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-slp-vectorize -Iinclude -Iaidadsp-lv2_amd/csrc $EXTRA \
   -S --cuda-device-only -o /tmp/st/convs.s aidadsp-lv2_amd/csrc/aidax_convs.hip
