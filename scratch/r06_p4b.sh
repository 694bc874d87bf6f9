#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu -k "not conv and not stack and not mfma and not extension" 2>&1 | tail -5
bash scratch/r06_ab_cfg2.sh r5=scratch/prev_lib/libaidax_r5_ship.so r6=aidadsp-lv2_amd/lib/libaidax_hip.so hooks=aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
