#!/bin/bash
# k_gru_gs: the model inputs as multiply-adds in the MFMAs' shadow (head) against the fp32 k-step (INM), one box, twice; parity of the head
for rep in 1 2; do
for v in "" INM; do
  echo "== build: ${v:-head}"
  if [ -n "$v" ]; then export AIDAX_LIB=scratch/prev_lib/libaidax_$v.so; else unset AIDAX_LIB; fi
  GS_ONLY=1 python scratch/gs_ab.py 2>&1 | grep GRU
done
done 2>&1
unset AIDAX_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gate_major or gru or cfg3" 2>&1 | tail -3
