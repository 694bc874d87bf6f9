#!/bin/bash
# A/B of AIDAX_TUNE settings on the bench's cfg2 region: scratch/ab.sh "0 1 2 3" [extra bench args]
for t in $1; do
  AIDAX_TUNE=$t python bench.py --no-others --no-cpu-baseline --steps 4000 --warmup 200 ${@:2} 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tune=$t', j['config']['kernel'], 'kernel_us=%.2f'%(j['roofline']['kernel_ms']*1e3), 'value=%.4g'%j['value'], 'err=%.2e'%(j['max_abs_err'] or 0))"
done
