#!/bin/bash
for rep in 1 2; do
for v in "" SE; do
  echo "== build: ${v:-head (stash at the end of the tick)}"
  if [ -n "$v" ]; then export AIDAX_LIB=scratch/prev_lib/libaidax_$v.so; else unset AIDAX_LIB; fi
  python scratch/ls_ab.py 2>&1 | grep "bf16x3"
done
done 2>&1
