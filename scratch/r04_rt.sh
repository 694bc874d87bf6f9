#!/bin/bash
mkdir -p gpurun_out
python scratch/rt_hist.py 20000 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_rt_hist.txt
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r04_rt -o r -- $(command -v python3) $root/scratch/rt_hist.py 5000 > $root/gpurun_out/prof_r04_rt.log 2>&1
cd $root
cat gpurun_out/r04_rt_hist.txt
grep "k_lstm_pipe" gpurun_out/prof_r04_rt/r_kernel_stats.csv
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1])
for o in j['other_workloads']: print(o['workload'][:5], o['kernel'], o['roofline'].get('traffic'), o['roofline_compute'].get('mfma_busy_frac'), o['roofline'].get('traffic_source','')[:60])
"
