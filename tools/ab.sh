#!/bin/bash
# A/B of builds of libmvba.so on the SAME box: every tools/ab/libmvba_*.so and the in-tree build ("tree"),
# alternating, kernel times from bench.py --no-cpu-baseline.   usage: tools/ab.sh [rounds]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for r in $(seq 1 ${1:-3}); do
  for lib in tools/ab/libmvba_*.so tree; do
    if [ $lib = tree ]; then unset MVBA_LIBRARY; v=tree; else export MVBA_LIBRARY=$PWD/$lib; v=$(basename $lib .so); fi
    timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -3 gpurun_out/ab_$v.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); k=d['kernel_ms_per_step']; print('$v'.ljust(14), round(d['value'],1), 'schur', round(k['schur'],3), 'solve', round(k['solve'],3), 'backsub', round(k['backsub_cost'],3), 'k1', round(k['resid_jac'],3), 'inv', round(k['point_inv'],3))"
  done
done
