#!/bin/bash
# quick A/B of the Schur kernel: parity subset, bench line, FETCH_SIZE / TCC hit-miss of k_schur_pairs
# usage: tools/quick_pairs.sh <tag> [env assignments...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_linearisation or random_scene or extreme or config4_shape" > gpurun_out/${tag}_t.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${tag}_t.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/${tag}_b.json 2> gpurun_out/${tag}_b.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/${tag}_b.json')); print('${tag}', round(d['value'],1), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${tag} -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pmc_${tag}/ | grep -A4 "^k_schur_"
