#!/bin/bash
# where the dense-visibility Schur kernel's time goes: timing-only builds with parts knocked out (-DMVBA_DENSE_KO bits: 1 main MFMAs,
# 2 columns of J~ and rows of G, 4 per-observation phase, 8 per-camera MFMAs; 9 = producers only, 6 = consumers only), K3 at 1 M points x 12 cameras
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 200 python tools/time_schur.py 1000000 12 1.0 || exit 1
for k in 9 6 1 8; do
  MVBA_LIBRARY=$PWD/tools/ab/libmvba_dko$k.so timeout -k 10 200 python tools/time_schur.py 1000000 12 1.0 || exit 1
done
