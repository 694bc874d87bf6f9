#!/bin/bash
# N probe calls in a row, each on whatever box the pool hands out: scratch/r05_probe_loop.sh N
cd /root/repo
for i in $(seq 1 ${1:-4}); do
  gpurun --timeout 300 -- "mkdir -p gpurun_out/r05; python scratch/r05_probe.py 300 > gpurun_out/r05/probe_\$(date +%H%M%S).txt 2>&1; cat gpurun_out/r05/probe_*.txt | tail -30" 2>&1 | tail -34
done
