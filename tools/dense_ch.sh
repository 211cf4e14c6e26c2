#!/bin/bash
# dense-visibility K3: chunk size / workgroups per CU.  HISTORICAL: the macros -DMVBA_DENSE_CH / -DMVBA_DENSE_WGS existed for this experiment only; the result is
# dense_ch() / dense_wgs() in csrc/mvba.hip (two 8-wave workgroups of 4-point chunks per CU up to 7 tiles), numbers in profiles/r05_dense_form.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "1000000 12 1.0" "1000000 8 1.0" "2000000 6 1.0" "1000000 14 1.0" "1000000 12 0.8"; do
  for lib in ch8w1 ch4w2 ch4w3; do
    MVBA_LIBRARY=$PWD/tools/ab/libmvba_$lib.so timeout -k 10 200 python tools/time_schur.py $shape || exit 1
  done
done
