#!/bin/bash
mkdir -p gpurun_out
for v in "" NOCELL NOMFMA NOPUBLISH; do
  echo "== build: ${v:-head}"
  if [ -n "$v" ]; then export AIDAX_LIB=scratch/prev_lib/libaidax_$v.so; else unset AIDAX_LIB; fi
  LS_ONLY=1 python scratch/ls_ab.py 2>&1 | grep "bf16x3"
done 2>&1 | tee gpurun_out/r04_ls_parts.txt
