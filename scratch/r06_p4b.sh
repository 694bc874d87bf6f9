#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_streams or full_size" 2>&1 | tail -5
bash scratch/r06_ab_cfg2.sh r5=scratch/prev_lib/libaidax_r5_ship.so r6=aidadsp-lv2_amd/lib/libaidax_hip.so hooks=aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
