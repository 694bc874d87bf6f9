#!/bin/bash
# cfg4 on the hooks build with AIDAX_TUNE settings: scratch/r05_cfg4_tune.sh "0 512 ..."
cd "$(dirname "$0")/.."
for t in $1; do
  echo -n "AIDAX_TUNE=$t: "
  AIDAX_TUNE=$t AIDAX_LIB=aidadsp-lv2_amd/lib/hooks/libaidax_hip.so python bench.py --workload cfg4 --no-others --no-cpu-baseline --no-traffic --no-dist --steps 2000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], round(d['ms_per_step']*1e3,2), 'us', d['max_abs_err'])"
done
