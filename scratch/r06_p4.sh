#!/bin/bash
mkdir -p gpurun_out/r06_p4
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_streams or lstm32 or cfg2 or reference_variant" 2>&1 | tail -15 | tee gpurun_out/r06_p4/tests.txt
bash scratch/r06_ab_cfg2.sh r5=scratch/prev_lib/libaidax_r5_ship.so r6=aidadsp-lv2_amd/lib/libaidax_hip.so 2>&1 | tee gpurun_out/r06_p4/ab_cfg2.txt
