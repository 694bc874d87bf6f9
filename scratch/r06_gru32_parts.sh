#!/bin/bash
# k_gru_pipe4<32> (measurement build -DAIDAX_P4_ALL_CELLS) taken apart with the test build's tune bits: no chain passes (262144), no Dense (524288), both;
# no issue priority for the recurrent waves (1)
cd "$(dirname "$0")/.."
export CELLS_LIB=$PWD/scratch/prev_lib/p4all/libaidax_hip.so MID=1
for t in 0 262144 524288 786432 1; do echo "AIDAX_TUNE=$t"; AIDAX_TUNE=$t python scratch/r06_pipe4_cells.py | grep "gru-32\|lstm-24"; done
