#!/bin/bash
# HBM traffic of cfg4 per launch, k_conv_st against k_conv_ms (hooks build): bench.py's own PMC passes (FETCH_SIZE, WRITE_SIZE)
cd "$(dirname "$0")/.."
export AIDAX_LIB=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
for st in 1 0; do
  AIDAX_CONV_ST=$st python bench.py --workload cfg4 --no-others --no-cpu-baseline --no-dist --steps 500 --warmup 50 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ST=$st', d['config']['kernel'], round(d['ms_per_step']*1e3,2), 'us', 'traffic', r.get('traffic'), r.get('traffic_source'))"
done
