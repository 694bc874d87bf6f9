#!/bin/bash
mkdir -p gpurun_out
./scratch/uoverlap3 > gpurun_out/r04_overlap3.txt 2>&1
./scratch/uoverlap2 1 > gpurun_out/r04_overlap_1wg.txt 2>&1
cat gpurun_out/r04_overlap3.txt
cat gpurun_out/r04_overlap_1wg.txt
