#!/bin/bash
# sweep the slot-resident Schur kernel's schedule knobs: bench line + L2 hit/miss of k_schur_slots
# usage: tools/sweep_slots.sh "VAR=a VAR2=b" "VAR=c" ...   (one quoted group of assignments per run)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  ( for kv in $grp; do export "$kv"; done
    timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/sws_$i.json 2> gpurun_out/sws_$i.err || { tail -3 gpurun_out/sws_$i.err; exit 0; }
    python -c "
import json; d=json.load(open('gpurun_out/sws_$i.json')); r=d['roofline_schur']; print('$grp', round(d['value'],1), 'schur', round(d['kernel_ms_per_step']['schur'],3), 'rows/items', round((r['slot_rows_incl_padding'] or 0)/r['items'],3), 'create_s', round(d['config']['engine_create_s'],2))"
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_sws_$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sws_$i.log 2>&1 && python tools/pmc_summary.py gpurun_out/pmc_sws_$i/ | grep -A2 "^k_schur_[ps]" | tr '\n' ' '; echo )
done
