#!/bin/bash
# cfg2's kernel time on several builds of the library (AIDAX_LIB): scratch/r05_libs_cfg2.sh lib1 lib2 ...
cd "$(dirname "$0")/.."
for rep in 1 2; do for lib in "$@"; do
  echo -n "$lib: "; AIDAX_LIB=$lib python bench.py --workload ${WL:-cfg2} --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 3000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"
done; done
