#!/bin/bash
# dense-visibility K3: s_setprio of the producer waves (0 = none, 1 = the tree's, 2, 3), K3 per launch at a few shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "1000000 12 1.0" "1000000 8 1.0" "1000000 20 1.0" "1000000 12 0.8"; do
  MVBA_LIBRARY=$PWD/tools/ab/libmvba_prio0.so timeout -k 10 200 python tools/time_schur.py $shape || exit 1
  timeout -k 10 200 python tools/time_schur.py $shape || exit 1
  MVBA_LIBRARY=$PWD/tools/ab/libmvba_prio2.so timeout -k 10 200 python tools/time_schur.py $shape || exit 1
  MVBA_LIBRARY=$PWD/tools/ab/libmvba_prio3.so timeout -k 10 200 python tools/time_schur.py $shape || exit 1
done
timeout -k 10 100 python tools/dense_trace.py 1000000 12 1.0
