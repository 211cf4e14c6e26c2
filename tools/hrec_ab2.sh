#!/bin/bash
# does the pace of a range hang on its diagonal waves?  sub-lists of the diagonal pairs cut shorter (MVBA_SLOT_DIAG_SCALE)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
line() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 240 python bench.py --no-cpu-baseline --svd-rows 0 --depth-rows 0 "$@" > gpurun_out/hrec_$tag.json 2> gpurun_out/hrec_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/hrec_$tag.err; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/hrec_$tag.json')); k=d['kernel_ms_per_step']; r=d['roofline_schur']
print('$tag'.ljust(20), 'it/s', round(d['value'],1), r['kernel'], 'schur', round(k['schur']/max(d['inner_solves'],1)*d['steps'],3), 'ms/solve', 'rows/items', round((r.get('slot_rows_incl_padding') or 0)/r['items'],3))"
}
for sc in 1.0 1.2 1.4 1.7 2.0; do
  export MVBA_SLOT_DIAG_SCALE=$sc
  export MVBA_LIBRARY=$PWD/tools/ab/libmvba_hrec33.so; line c3_hrec33_d$sc --steps 10 --warmup 3
done
for sc in 1.0 1.2 1.4 1.7; do
  export MVBA_SLOT_DIAG_SCALE=$sc
  unset MVBA_LIBRARY; line m90_tree_d$sc --cams 90 --steps 10 --warmup 3
done
