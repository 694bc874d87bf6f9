#!/bin/bash
# cfg4 on the shipped library (and, with AIDAX_LIB, on a variant): kernel, us per block, max error against the oracle, per-launch distribution
cd "$(dirname "$0")/.."
python bench.py --workload cfg4 --no-others --no-cpu-baseline --no-traffic --steps 2000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], round(d['ms_per_step']*1e3,2), 'us', d['max_abs_err'], {k: round(v,1) for k,v in d['per_launch_us'].items()}, d['gpu_state_while_running'])"
