#!/bin/bash
# k_mfma_ls, the lower layer's head and tail (cfg5): A/B of measurement builds on one box, twice
for rep in 1 2; do
for v in "" TB SF XL ALL3; do
  echo "== build: ${v:-head}"
  if [ -n "$v" ]; then export AIDAX_LIB=scratch/prev_lib/libaidax_$v.so; else unset AIDAX_LIB; fi
  LS_ONLY=1 python scratch/ls_ab.py 2>&1 | grep "bf16x3"
done
done 2>&1
