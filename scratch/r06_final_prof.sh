#!/bin/bash
# round 6, final evidence: the GPU suite (with its ship leg), kernel stats + PMC passes for every BASELINE config, the bench line as the driver runs it
tag=${1:-r06}
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/${tag}_gpu_suite.txt
cat gpurun_out/${tag}_gpu_suite.txt
for cfg in cfg2 cfg3 cfg4 cfg5; do
  steps=2000; [ $cfg = cfg3 ] && steps=600; [ $cfg = cfg5 ] && steps=250
  bash profiles/run_profiles.sh ${tag}_$cfg $cfg $steps > gpurun_out/prof_${tag}_$cfg.log 2>&1
  python3 profiles/summarize.py gpurun_out/prof_${tag}_$cfg gpurun_out/${tag}_$cfg > /dev/null 2>&1
  head -3 gpurun_out/${tag}_${cfg}_kernel_stats.csv | cut -c1-160
done
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_style.json 2> gpurun_out/${tag}_bench_driver_style.err
tail -c 1200 gpurun_out/${tag}_bench_driver_style.json
