#!/bin/bash
# k_gru_pipe4<32> / <20> (measurement build -DAIDAX_P4_ALL_CELLS): recurrent wave j starts j x k x 16 cycles late (test build, bits 24 .. 27 of AIDAX_TUNE = k) —
# do the four waves of a workgroup stand in each other's way at the LDS when they run in step?
cd "$(dirname "$0")/.."
export CELLS_LIB=$PWD/scratch/prev_lib/p4all/libaidax_hip.so MID=1
for k in 0 1 2 3 4 6 8 12 15; do echo -n "k=$k  "; AIDAX_TUNE=$((k << 24)) python scratch/r06_pipe4_cells.py | grep "gru-32" | sed 's/  */ /g'; done
for k in 0 2 4 8; do echo -n "k=$k  "; AIDAX_TUNE=$((k << 24)) python scratch/r06_pipe4_cells.py | grep "gru-20" | sed 's/  */ /g'; done
