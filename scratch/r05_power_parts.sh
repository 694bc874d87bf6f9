#!/bin/bash
# cfg5's power and clock by parts: the measurement builds of k_mfma_ls (scratch/mkvariant.sh NAME -DAIDAX_LS_NAME: every one but the
# first computes wrong results) under the probe, which reads rocm-smi while the kernel runs.
cd "$(dirname "$0")/.."
echo "== head"; python scratch/r05_probe.py child cfg5 300 head watch
for v in NOSHIP NOMFMA NOCELL NOPUBLISH NOXW SKEL; do
  [ -f scratch/prev_lib/libaidax_$v.so ] || continue
  echo "== $v"; AIDAX_LIB=scratch/prev_lib/libaidax_$v.so python scratch/r05_probe.py child cfg5 300 $v watch
done
echo "== cfg3 (k_gru_gs)"; python scratch/r05_probe.py child cfg3 300 cfg3 watch
echo "== cfg2"; python scratch/r05_probe.py child cfg2 300 cfg2 watch
echo "== cfg4"; python scratch/r05_probe.py child cfg4 300 cfg4 watch
echo "== cfg5 on k_mfma_lp (fp32 MFMA)"; AIDAX_LP_SPLIT=0 python scratch/r05_probe.py child cfg5 300 lp watch
echo "== cfg5 on k_mfma (no ring)"; AIDAX_KERNEL=mfma python scratch/r05_probe.py child cfg5 300 mfma watch
