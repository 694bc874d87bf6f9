#!/bin/bash
# the GPU suite next to busy loops on every core (a crowded host): which tests lean on timing?
n=${1:-$(nproc)}
pids=""
for i in $(seq 1 $n); do ( while :; do :; done ) & pids="$pids $!"; done
timeout 1500 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -150 > gpurun_out/suite_busy.txt
for p in $pids; do kill $p; done
tail -15 gpurun_out/suite_busy.txt
