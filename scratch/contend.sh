#!/bin/bash
# timing-sensitive tests under host contention: pinned to two cores, with busy loops next to them
nproc > gpurun_out/contend.txt
for mode in ${CONTEND_MODES:-plain pinned2 pinned1 busy}; do
  echo "== $mode" >> gpurun_out/contend.txt
  pids=""
  if [ $mode = busy ]; then for i in $(seq 1 $(nproc)); do ( while :; do :; done ) & pids="$pids $!"; done; fi
  for rep in 1 2 3; do
    case $mode in
      pinned2) pre="taskset -c 0,1" ;;
      pinned1) pre="taskset -c 0" ;;
      *) pre="" ;;
    esac
    $pre timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_hub.py -q -m gpu --tb=line -k "worker_thread_prepares or deadline_that_splits" 2>&1 | tail -4 >> gpurun_out/contend.txt
  done
  for p in $pids; do kill $p; done
done
cat gpurun_out/contend.txt
