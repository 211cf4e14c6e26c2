#!/bin/bash
# Gram + Jacobi route against the block iteration at 33..64 columns (MVSVD_WIDE_MIN), one-shot factorisation and the depth loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for wm in 64 32; do
  echo "== MVSVD_WIDE_MIN=$wm"
  MVSVD_WIDE_MIN=$wm MVBA_SVD_CACHE=0 timeout -k 10 200 python tools/time_svd_wide.py 1000000 36 48 64 --reps 2 || exit 1
  MVSVD_WIDE_MIN=$wm timeout -k 10 200 python tools/time_depth_step.py 1000000 12 || exit 1
done
