cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
for seed in 91 92 93; do export SOAK_SEED=$seed
  SOAK_STREAMS=1 python tests/soak.py 600 2>&1 | tail -1
  SOAK_STREAMS=3 python tests/soak.py 400 2>&1 | tail -1
  AIDAX_KERNEL_WORD=0 SOAK_STREAMS=1 python tests/soak.py 300 2>&1 | tail -1
done
