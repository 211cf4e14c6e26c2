#!/bin/bash
# sweep schedule knobs of the Schur kernel on config 4's per-GPU shard (1.25 M points x 500 cameras x 5 %):
# bench line (Schur ms per solve, step rows / items) and, with PMC=1 in a group, the L2 hit / miss counts of the kernel
# usage: tools/sweep_c4.sh "VAR=a VAR2=b" "VAR=c PMC=1" ...   (one quoted group of assignments per run)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ARGS="--points 1250000 --cams 500 --vis 0.05 --steps 4 --warmup 1 --no-cpu-baseline --svd-rows 0 --depth-rows 0"
i=0
for grp in "$@"; do
  i=$((i+1))
  ( for kv in $grp; do export "$kv"; done
    timeout -k 10 300 python bench.py $ARGS > gpurun_out/swc_$i.json 2> gpurun_out/swc_$i.err || { echo "$grp | FAILED:"; tail -2 gpurun_out/swc_$i.err; exit 0; }
    python -c "
import json; d=json.load(open('gpurun_out/swc_$i.json')); r=d['roofline_schur']; print('$grp', '|', r['kernel'], 'schur', round(d['kernel_ms_per_step']['schur']/max(d['inner_solves'],1)*d['steps'],3), 'ms/solve  step', round(d['ms_per_step'],2), 'rows/items', round((r['slot_rows_incl_padding'] or 0)/r['items'],3), 'create_s', round(d['config']['engine_create_s'],2))"
    if [ -n "$PMC" ]; then
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmc_swc_$i -- python bench.py --points 1250000 --cams 500 --vis 0.05 --steps 2 --warmup 1 --no-cpu-baseline --svd-rows 0 > gpurun_out/pmc_swc_$i.log 2>&1 && python tools/pmc_summary.py gpurun_out/pmc_swc_$i/ | grep -A3 "^k_schur_[ps]" | tr '\n' ' '; echo
    fi )
done
