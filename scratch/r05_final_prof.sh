#!/bin/bash
# round 5, final evidence: kernel stats + PMC passes for every BASELINE config, the bench line as the driver runs it, the one-stream latency
tag=${1:-r05}
for cfg in cfg2 cfg3 cfg4 cfg5; do
  steps=2000; [ $cfg = cfg3 ] && steps=600; [ $cfg = cfg5 ] && steps=250
  bash profiles/run_profiles.sh ${tag}_$cfg $cfg $steps > gpurun_out/prof_${tag}_$cfg.log 2>&1
  python3 profiles/summarize.py gpurun_out/prof_${tag}_$cfg gpurun_out/${tag}_$cfg > /dev/null 2>&1
  head -3 gpurun_out/${tag}_${cfg}_kernel_stats.csv | cut -c1-160
done
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_style.json 2> gpurun_out/${tag}_bench_driver_style.err
tail -c 1500 gpurun_out/${tag}_bench_driver_style.json
