#!/bin/bash
# bench line of a scene shape under groups of environment settings (Schur ms per solve, step rows / items; PMC=1 in a
# group adds the L2 hit / miss counts of the Schur kernel)
# usage: tools/sweep_shape.sh "--points N --cams M --vis P" "VAR=a VAR2=b" "VAR=c PMC=1" ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SHAPE="$1"; shift
ARGS="$SHAPE --steps 4 --warmup 1 --no-cpu-baseline --svd-rows 0"
i=0
for grp in "$@"; do
  i=$((i+1))
  ( for kv in $grp; do export "$kv"; done
    timeout -k 10 300 python bench.py $ARGS > gpurun_out/sws_$i.json 2> gpurun_out/sws_$i.err || { echo "$grp | FAILED"; tail -2 gpurun_out/sws_$i.err; exit 0; }
    python -c "
import json; d=json.load(open('gpurun_out/sws_$i.json')); r=d['roofline_schur']; print('$SHAPE |', '$grp', '|', r['kernel'], 'schur', round(d['kernel_ms_per_step']['schur']/max(d['inner_solves'],1)*d['steps'],3), 'ms/solve  step', round(d['ms_per_step'],2), 'rows/items', round((r['slot_rows_incl_padding'] or 0)/r['items'],3), 'items', r['items'], 'create_s', round(d['config']['engine_create_s'],2))"
    if [ -n "$PMC" ]; then
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmc_sws_$i -- python bench.py $SHAPE --steps 2 --warmup 1 --no-cpu-baseline --svd-rows 0 > gpurun_out/pmc_sws_$i.log 2>&1 && python tools/pmc_summary.py gpurun_out/pmc_sws_$i/ | grep -A3 "^k_schur_[ps]" | tr '\n' ' '; echo
    fi )
done
