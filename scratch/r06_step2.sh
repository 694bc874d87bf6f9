#!/bin/bash
mkdir -p gpurun_out/r06_s2
bash scratch/r06_ab_cfg2.sh r5=scratch/prev_lib/libaidax_r5_ship.so r6=aidadsp-lv2_amd/lib/libaidax_hip.so 2>&1 | tee gpurun_out/r06_s2/ab_cfg2.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r06_s2/suite.txt
cat gpurun_out/r06_s2/suite.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_s2/bench.json 2> gpurun_out/r06_s2/bench.err
tail -3 gpurun_out/r06_s2/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_s2/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['kernel'], d['max_abs_err'])
for o in d['other_workloads']: print(o['kernel'], round(o['ms_per_step']*1e3,1), o['max_abs_err'])
for f in d['host_inclusive']['forms']: print(round(f['us_per_block'],1), f['form'])
def show(cs):
    for c in cs:
        print(c['case'][:44], c['frames'], 'paced', {k:round(v,1) for k,v in c['paced_us'].items() if k!='calls'}, 'b2b', {k:round(v,1) for k,v in c['back_to_back_us'].items() if k!='calls'})
        if 'kernel_us_paced' in c: print('    kernel paced', {k:round(v,1) for k,v in c['kernel_us_paced'].items() if k!='calls'}, 'b2b', {k:round(v,1) for k,v in c['kernel_us_back_to_back'].items() if k!='calls'})
show(d['realtime_paced']['cases']); print('keep warm:'); show(d['realtime_paced']['keep_warm']['cases'])
PY
