#!/bin/bash
# the hand-over ring at the XCD's L2 (sc0) instead of device scope (sc0 sc1): time and parity of the stacked kernels
for rep in 1 2; do
for v in "" L2; do
  echo "== build: ${v:-head}"
  if [ -n "$v" ]; then export AIDAX_LIB=scratch/prev_lib/libaidax_$v.so; else unset AIDAX_LIB; fi
  python scratch/ls_ab.py 2>&1 | grep "bf16x3\|fp32"
done
done 2>&1
export AIDAX_LIB=scratch/prev_lib/libaidax_L2.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lp_tenancy.py -x -q -m gpu -k "split_stack or layer_pipelined or stacked or full_size or tenancy or drift_lstm96" 2>&1 | tail -5
