#!/bin/bash
mkdir -p gpurun_out/r06_sb
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_sb/suite.txt
cat gpurun_out/r06_sb/suite.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_sb/bench.json 2> gpurun_out/r06_sb/bench.err
tail -c 6000 gpurun_out/r06_sb/bench.json
tail -5 gpurun_out/r06_sb/bench.err
