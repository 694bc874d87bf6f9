#!/bin/bash
# cfg2 through two or more builds of the library on ONE box, interleaved, three rounds: name=path ...
cd "$(dirname "$0")/.."
run() { AIDAX_LIB=$2 python bench.py --workload cfg5 --no-others --no-cpu-baseline --no-traffic --no-dist --steps 300 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us', d['max_abs_err'])"; }
for round in 1 2 3; do for v in "$@"; do run "${v%%=*}" "$PWD/${v#*=}"; done; done
