#!/bin/bash
# round 6: the GPU suite with its ship leg, then the hooks leg once more (digests only) to see which tests' outputs are reproducible at all
mkdir -p gpurun_out/r06_ship
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_ship/suite.txt
AIDAX_DIGEST_OUT=gpurun_out/r06_ship/hooks2.json python -m pytest tests -q -m gpu --deselect tests/test_gpu_ship.py 2>&1 | tail -5 > gpurun_out/r06_ship/hooks2.txt
AIDAX_SHIP_LEG=1 AIDAX_DIGEST_OUT=gpurun_out/r06_ship/ship2.json python -m pytest tests -q -m gpu -rs 2>&1 | tail -80 > gpurun_out/r06_ship/ship2.txt
AIDAX_DIGEST_OUT=gpurun_out/r06_ship/hooks3.json python -m pytest tests -q -m gpu --deselect tests/test_gpu_ship.py 2>&1 | tail -5 > gpurun_out/r06_ship/hooks3.txt
cat gpurun_out/r06_ship/suite.txt
