#!/bin/bash
# dense-visibility K3: four against eight consumer waves up to 8 tiles (-DMVBA_DENSE_NC_SMALL=8: 16 waves per workgroup, two MFMA-issuing waves per SIMD)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "1000000 12" "1000000 8" "2000000 6" "1000000 14"; do
  timeout -k 10 200 python tools/time_schur.py $shape 1.0 || exit 1
  MVBA_LIBRARY=$PWD/tools/ab/libmvba_nc8.so timeout -k 10 200 python tools/time_schur.py $shape 1.0 || exit 1
done
