#!/bin/bash
# the shipped library against the hooks build on the bench workloads (same sources; the hooks build keeps the kernels' tune tests)
cd "$(dirname "$0")/.."
for wl in ${1:-cfg2}; do for lib in aidadsp-lv2_amd/lib/libaidax_hip.so aidadsp-lv2_amd/lib/hooks/libaidax_hip.so aidadsp-lv2_amd/lib/libaidax_hip.so aidadsp-lv2_amd/lib/hooks/libaidax_hip.so; do
  echo -n "$wl $lib: "; AIDAX_LIB=$lib python bench.py --workload $wl --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 3000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us kernel,', round(d['ms_per_step']*1e3,2), 'us per step')"
done; done
