#!/bin/bash
# cfg2 through builds of the library on ONE box, interleaved, three rounds: name=path ...
cd "$(dirname "$0")/.."
run() { AIDAX_LIB=$2 python bench.py --workload cfg2 --no-others --no-cpu-baseline --no-traffic --no-dist --steps 4000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us', d['max_abs_err'])"; }
for round in 1 2; do for v in "$@"; do run "${v%%=*}" "$PWD/${v#*=}"; done; done
