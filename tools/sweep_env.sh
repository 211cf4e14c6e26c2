#!/bin/bash
# bench line (it/s + per-kernel ms) for each value of one environment knob
# usage: tools/sweep_env.sh VAR v1 v2 ...
cd $GRAFT_REPO_ROOT
var=$1; shift
for v in "$@"; do
  export $var=$v
  timeout -k 10 200 python bench.py --no-cpu-baseline 2> gpurun_out/sweep_env.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$var=$v', round(d['value'],1), ' '.join('%s %.3f' % (n, k[n]) for n in ('resid_jac','point_inv','schur','solve','backsub_cost')))"
done
