#!/bin/bash
# A/B of the h-in-the-record timing builds (tools/build_hrec_timing.sh) against the product library on one box:
# config 3 (tree vs hrec33) and 1 M x 94 cameras (one round also at 8 waves per CU: tree, hrec33, hrec32, hrec42).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
line() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 240 python bench.py --no-cpu-baseline --svd-rows 0 --depth-rows 0 "$@" > gpurun_out/hrec_$tag.json 2> gpurun_out/hrec_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/hrec_$tag.err; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/hrec_$tag.json')); k=d['kernel_ms_per_step']; r=d['roofline_schur']
print('$tag'.ljust(16), 'it/s', round(d['value'],1), r['kernel'], 'schur', round(k['schur']/max(d['inner_solves'],1)*d['steps'],3), 'ms/solve', 'k1', round(k['resid_jac'],3), 'inv', round(k['point_inv'],3), 'rows/items', round((r.get('slot_rows_incl_padding') or 0)/r['items'],3))"
}
for rep in 1 2; do
  unset MVBA_LIBRARY; line c3_tree_$rep --steps 20 --warmup 5
  export MVBA_LIBRARY=$PWD/tools/ab/libmvba_hrec33.so; line c3_hrec33_$rep --steps 20 --warmup 5
done
for v in tree hrec33 hrec32 hrec42; do
  if [ $v = tree ]; then unset MVBA_LIBRARY; elif [ $v = hrec33 ]; then export MVBA_LIBRARY=$PWD/tools/ab/libmvba_$v.so; else export MVBA_LIBRARY=$PWD/tools/ab2/libmvba_$v.so; fi
  line m94_$v --cams 94 --steps 20 --warmup 5
done
