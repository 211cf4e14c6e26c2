#!/bin/bash
# sweep an environment knob of the pair-major Schur kernel: bench line + L2 hit/miss of k_schur_pairs
# usage: tools/sweep_pairs.sh VAR v1 v2 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
var=$1; shift
for v in "$@"; do
  export $var=$v
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/sw_${var}_$v.json 2> gpurun_out/sw_${var}_$v.err || { tail -3 gpurun_out/sw_${var}_$v.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/sw_${var}_$v.json')); print('$var=$v', round(d['value'],1), 'schur', round(d['kernel_ms_per_step']['schur'],3), 'units', d['roofline_schur']['units'])"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_sw_${var}_$v -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sw_${var}_$v.log 2>&1 && python tools/pmc_summary.py gpurun_out/pmc_sw_${var}_$v/ | grep -A2 "^k_schur_[ps]" | tr '\n' ' '; echo
done
