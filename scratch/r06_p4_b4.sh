#!/bin/bash
# k_*_pipe4 with hand-overs every 4 frames (four macro steps per tile) against every 8 (two): parity, then the fixed cost and cfg2
cd "$(dirname "$0")/.."
B8=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so; B4=$PWD/scratch/prev_lib/p4b4/libaidax_hip.so
AIDAX_LIB=$B4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_streams" 2>&1 | tail -2
for r in 1 2; do
  echo -n "B=8 "; AIDAX_LIB=$B8 python scratch/r06_fixed_cost.py | head -1
  echo -n "B=4 "; AIDAX_LIB=$B4 python scratch/r06_fixed_cost.py | head -1
done
echo -n "B=8 "; AIDAX_LIB=$B8 python scratch/r06_fixed_cost.py lstm 16 1024 | head -1
echo -n "B=4 "; AIDAX_LIB=$B4 python scratch/r06_fixed_cost.py lstm 16 1024 | head -1
echo -n "B=8 "; AIDAX_LIB=$B8 python scratch/r06_fixed_cost.py gru 12 1024 | head -1
echo -n "B=4 "; AIDAX_LIB=$B4 python scratch/r06_fixed_cost.py gru 12 1024 | head -1
bash scratch/r06_ab_cfg2.sh b8=aidadsp-lv2_amd/lib/hooks/libaidax_hip.so b4=scratch/prev_lib/p4b4/libaidax_hip.so
