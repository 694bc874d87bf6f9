#!/bin/bash
# the hand-over ring at other cache scopes / depths (scratch/mkvariant_lp.sh builds): time, power, and whether the stacked-model tests still pass
cd "$(dirname "$0")/.."
T="tests/test_gpu_parity.py -k split_stack_geometries or random_recurrent_stacks or stacked_one_launch or several_ranges or layer_pipelined_kernel_keeps"
echo "== head"; python scratch/r05_probe.py child cfg5 300 head watch | grep -v "^ *device\|while running: device"
for v in "$@"; do
  [ -f scratch/prev_lib/libaidax_$v.so ] || continue
  echo "== $v"; AIDAX_LIB=scratch/prev_lib/libaidax_$v.so python scratch/r05_probe.py child cfg5 300 $v watch | grep -v "while running: device"
  AIDAX_LIB=$PWD/scratch/prev_lib/libaidax_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_stack_geometries or random_recurrent_stacks or stacked_one_launch or several_ranges or layer_pipelined_kernel_keeps or long_run_drift_lstm96x2_on_the_layer" 2>&1 | tail -3
done
