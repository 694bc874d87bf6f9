#!/bin/bash
# k_mfma_ls / k_mfma_ls1: the first layer's bias and input reads requested ahead of the fragment reads (head) against behind them (BL), one box, twice
for rep in 1 2; do
for v in "" BL; do
  echo "== build: ${v:-head}"
  if [ -n "$v" ]; then export AIDAX_LIB=scratch/prev_lib/libaidax_$v.so; else unset AIDAX_LIB; fi
  python scratch/ls_ab.py 2>&1 | grep "bf16x3"
  AIDAX_LSTM_GS=0 LS1_SHAPES=lstm-64,lstm-80 python scratch/ls1_ab.py 2>&1 | grep "S=  4096" | sed 's/.*| ls1=/ls1=/'
done
done 2>&1
unset AIDAX_LIB
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_stack or lone_layer or layer_pipelined or stacked_one_launch or full_size" 2>&1 | tail -3
